#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03_alltests
for i in 1 2; do
  timeout 3300 python -m pytest tests -q -m gpu > gpurun_out/r03_alltests/pytest_$i.log 2>&1
  tail -4 gpurun_out/r03_alltests/pytest_$i.log
done
python tools/debug/member_mismatch.py human_k31 300000000 2>&1 | tail -9
