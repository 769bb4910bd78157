#!/bin/bash
# the bench harness tests with the fifth child (k = 31 streaming query, 25-base table key), then the driver's command
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_last; mkdir -p $out
timeout 2400 python -m pytest tests/test_gpu_bench_harness.py -x -q -m gpu 2>&1 | tail -4 | tee $out/pytest.txt
bash tools/jobs/r04_driver_command.sh 2>&1 | tail -12 | tee $out/driver_command.txt
