#!/bin/bash
# the streaming kernel after a change: its parity tests, then the three read sets of RESULTS.md on one box
cd "$(dirname "$0")/../.."
out=gpurun_out/${1:-r05_stream_check}; mkdir -p $out
( time timeout 1500 python -m pytest tests/test_gpu_streaming.py tests/test_gpu_km_sweep.py tests/test_gpu_baseline_workloads.py -x -q -m gpu -k "not full_size_dictionary" ) 2>&1 | tail -8 | tee $out/pytest_streaming.txt
bash tools/jobs/r05_stream_sets.sh ${1:-r05_stream_check}_bench 2>&1 | tail -3
