#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03_overlap
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r03_overlap/pytest_parity.log 2>&1
tail -3 gpurun_out/r03_overlap/pytest_parity.log
bash tools/jobs/r03_env_ab.sh r03_overlap SSHASH_AMD_OVERLAP 0 1 3
bash tools/jobs/r03_env_ab.sh r03_overlap_c2 SSHASH_AMD_OVERLAP 0 1 2 --workload c2
