#!/bin/bash
# the streaming kernel with the sliding election: tests, then kernel stats + PMC on the high-hit and the low-hit read set (k = 31)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04_roll_prof
timeout 1200 python -m pytest tests/test_gpu_streaming.py -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r04_roll_prof/pytest.txt
bash tools/jobs/r04_profile.sh r04_prof_stream_k31_high_rolling --workload c3 --bases 1000000000 --streaming --reads 20000000 --positive 0.95 --stream-oracle-reads 20000 2>&1 | tail -12
bash tools/jobs/r04_profile.sh r04_prof_stream_k31_low_rolling --workload c3 --bases 1000000000 --streaming --reads 20000000 --positive 0.0 --stream-oracle-reads 20000 2>&1 | tail -12
