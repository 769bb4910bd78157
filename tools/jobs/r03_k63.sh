#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03_k63
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r03_k63/pytest.log 2>&1
tail -4 gpurun_out/r03_k63/pytest.log
bash tools/jobs/r03_env_ab.sh r03_k63 SSHASH_AMD_INWAVE 0 1 3 --k 63 --m 25
