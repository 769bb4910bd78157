#!/bin/bash
# same-box A/B: the election with v_mad_u32_u24 + switch (new) against the previous commit (old); C4 (k = 63) and C3
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_ab_election; mkdir -p $out
ROUNDS=3 bash tools/jobs/r02_ab.sh --workload c4 --no-file-query --no-other-paths --no-other-workloads --steps 10 --warmup 2 2>&1 | tee $out/c4.txt
ROUNDS=3 bash tools/jobs/r02_ab.sh --workload c3 --no-file-query --no-other-paths --no-other-workloads --steps 10 --warmup 2 2>&1 | tee $out/c3.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_km_sweep.py -x -q -m gpu 2>&1 | tail -4 | tee $out/pytest.txt
