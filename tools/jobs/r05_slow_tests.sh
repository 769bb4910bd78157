#!/bin/bash
cd "$(dirname "$0")/../.."
out=gpurun_out/r05_slow_tests; mkdir -p $out
export TMPDIR=/tmp
( time timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python -m pytest tests/test_gpu_streaming.py -x -q -m gpu -k "test_streaming_counters_match_oracle and (se_regular or k63_regular)" ) 2>&1 | tail -6
find $out/trace -name 't_kernel_stats.csv' -exec cp {} $out/kernel_stats.csv \;
head -12 $out/kernel_stats.csv | cut -c1-260
rm -rf $out/trace
