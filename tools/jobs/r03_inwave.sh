#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03_inwave
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r03_inwave/pytest.log 2>&1
tail -4 gpurun_out/r03_inwave/pytest.log
bash tools/jobs/r03_env_ab.sh r03_inwave SSHASH_AMD_INWAVE 0 1 3
bash tools/jobs/r03_env_ab.sh r03_inwave_c2 SSHASH_AMD_INWAVE 0 1 3 --workload c2
