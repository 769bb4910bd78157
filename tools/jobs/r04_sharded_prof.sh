#!/bin/bash
# Round 4: where the time of the sharded lookup goes at N = 1 (route, exchange, lookup, return, combine)
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r04_sharded${TAG:-}
mkdir -p $OUT
export TMPDIR=/tmp
for mode in ${MODES:-table minimizer}; do
  B="python bench.py --workload c2 --sharded $mode --no-cpu-baseline --no-extra-mixes --steps 5 --warmup 2"
  python bench.py --workload c2 --sharded $mode --no-cpu-baseline --no-extra-mixes --steps 20 --warmup 3 > $OUT/bench_$mode.jsonl 2> $OUT/bench_$mode.err
  cut -c1-160 $OUT/bench_$mode.jsonl
  timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT/trace_$mode -o t -- $B > $OUT/trace_$mode.log 2>&1
  for n in kernel_stats memory_copy_stats; do
    f=$(find $OUT/trace_$mode -name "t_$n.csv" | head -1)
    [ -n "$f" ] && cp "$f" $OUT/${mode}_$n.csv && cut -c1-170 "$f" | head -14
  done
  rm -rf $OUT/trace_$mode
done
