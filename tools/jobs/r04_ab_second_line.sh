#!/bin/bash
# same-box A/B: k <= 63 first pass, slot 1's line fetched by ranked quads (new) against four more rounds of the whole wave (old); then the PMC profile of the new build
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_ab_second_line; mkdir -p $out
ROUNDS=3 bash tools/jobs/r02_ab.sh --workload c4 --no-file-query --no-other-paths --no-other-workloads --steps 10 --warmup 2 2>&1 | tee $out/c4.txt
timeout 600 python -m pytest tests/test_gpu_km_sweep.py tests/test_gpu_reference_data.py -x -q -m gpu 2>&1 | tail -4 | tee $out/pytest.txt
bash tools/jobs/r04_profile.sh r04_prof_c4 --workload c4 2>&1 | tail -40 | tee $out/profile_c4.txt
