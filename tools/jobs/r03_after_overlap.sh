#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03_after_overlap
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r03_after_overlap/pytest.log 2>&1; tail -3 gpurun_out/r03_after_overlap/pytest.log
bash tools/jobs/r03_final.sh
