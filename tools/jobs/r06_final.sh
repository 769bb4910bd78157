#!/bin/bash
# round 6: the driver's command, then the PMC records of the lines whose kernels changed + the headline
cd "$(dirname "$0")/../.."
export SSHASH_BENCH_CACHE=/tmp
bash tools/jobs/r06_driver_command.sh ${1:-r06_driver_command}
bash tools/jobs/r06_records.sh ${2:-}
