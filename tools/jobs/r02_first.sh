#!/bin/bash
# Round 2: bucketed table + quad-cooperative line fetch -- parity tests, then the C2 and C3 benches.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r02_first
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 2>&1 | tail -15
timeout 900 python bench.py --workload c2 --no-cpu-baseline > $OUT/bench_c2.jsonl 2> $OUT/bench_c2.err; tail -3 $OUT/bench_c2.err; cut -c1-1500 $OUT/bench_c2.jsonl
timeout 1500 python bench.py > $OUT/bench_c3.jsonl 2> $OUT/bench_c3.err; tail -8 $OUT/bench_c3.err; cat $OUT/bench_c3.jsonl
