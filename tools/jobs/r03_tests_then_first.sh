#!/bin/bash
# GPU parity tests that touch the replica layout, then the r03_first measurement
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/$1
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/$1/pytest_parity.log 2>&1
tail -5 gpurun_out/$1/pytest_parity.log
bash tools/jobs/r03_first.sh "$@"
