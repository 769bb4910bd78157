#!/bin/bash
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_ab_hop; mkdir -p $out
ROUNDS=4 bash tools/jobs/r02_ab.sh --workload c4 --no-file-query --no-other-paths --no-other-workloads --steps 10 --warmup 2 2>&1 | tee $out/c4.txt
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_km_sweep.py tests/test_gpu_reference_data.py tests/test_gpu_baseline_workloads.py tests/test_gpu_switches.py -x -q -m gpu ) 2>&1 | tail -5 | tee $out/pytest.txt
