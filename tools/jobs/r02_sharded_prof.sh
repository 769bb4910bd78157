#!/bin/bash
# Round 2: where the time of the sharded lookup goes at N = 1 (route, exchange, lookup, return, combine)
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r02_sharded
mkdir -p $OUT
export TMPDIR=/tmp
for mode in ${MODES:-table minimizer}; do
  B="python bench.py --workload c2 --sharded $mode --no-cpu-baseline --no-extra-mixes --steps 5 --warmup 2"
  $B > $OUT/bench_$mode.jsonl 2> $OUT/bench_$mode.err
  cut -c1-160 $OUT/bench_$mode.jsonl
  timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT/trace_$mode -o t -- $B > $OUT/trace_$mode.log 2>&1
  f=$(find $OUT/trace_$mode -name 't_kernel_stats.csv' | head -1)
  [ -n "$f" ] && cut -c1-150 "$f" | head -14
  f=$(find $OUT/trace_$mode -name 't_memory_copy_stats.csv' | head -1)
  [ -n "$f" ] && head -6 "$f"
done
