#!/bin/bash
cd "$(dirname "$0")/../.."
bash tools/jobs/r05_profile.sh r05_prof_stream_random --workload c3 --streaming --positive 0.0 --reads 20000000 2>&1 | tail -40
bash tools/jobs/r05_profile.sh r05_prof_stream_high --workload c3 --streaming --positive 0.95 --reads 20000000 2>&1 | tail -40
