#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03_inline
SSHASH_AMD_INLINE_RESUME=1 timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r03_inline/pytest_parity_inline.log 2>&1
tail -3 gpurun_out/r03_inline/pytest_parity_inline.log
export SSHASH_AMD_OVERLAP=0
bash tools/jobs/r03_env_ab.sh r03_inline SSHASH_AMD_INLINE_RESUME 0 1 3
unset SSHASH_AMD_OVERLAP
bash tools/jobs/r03_env_ab.sh r03_overlap_c2 SSHASH_AMD_OVERLAP 0 1 3 --workload c2
SSHASH_AMD_OVERLAP=0 bash tools/jobs/r03_env_ab.sh r03_inline_c2 SSHASH_AMD_INLINE_RESUME 0 1 2 --workload c2
