#!/bin/bash
cd "$(dirname "$0")/../.."
out=gpurun_out/r05_first; mkdir -p $out
( time timeout 2400 python -m pytest tests/test_gpu_bench_harness.py -x -q -m gpu ) 2>&1 | tail -15 | tee $out/pytest_harness.txt
bash tools/jobs/r05_driver_command.sh r05_first 2>&1 | tail -12
