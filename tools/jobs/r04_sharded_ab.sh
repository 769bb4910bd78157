#!/bin/bash
# Round 4: sharded lookup at N = 1 after the owners are elected once and the one-reply combine; tests first
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r04_sharded${TAG:-_b}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_sharded.py tests/test_gpu_bench_harness.py -m gpu -x -q > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
for mode in table minimizer; do
  python bench.py --workload c2 --sharded $mode --no-cpu-baseline --no-extra-mixes --steps 20 --warmup 3 > $OUT/bench_$mode.jsonl 2> $OUT/bench_$mode.err
  cut -c1-160 $OUT/bench_$mode.jsonl
done
B="python bench.py --workload c2 --sharded table --no-cpu-baseline --no-extra-mixes --steps 5 --warmup 2"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $B > $OUT/trace.log 2>&1
f=$(find $OUT/trace -name "t_kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $OUT/table_kernel_stats.csv && grep -i "route\|rccl\|fast_lookup\|fill" "$f" | cut -c1-60,150-260
rm -rf $OUT/trace
