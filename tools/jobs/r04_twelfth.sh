#!/bin/bash
cd "$(dirname "$0")/../.."
bash tools/jobs/r04_driver_command.sh 2>&1 | tail -12
bash tools/jobs/r04_final_profiles.sh 2>&1 | tail -14
