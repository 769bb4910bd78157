#!/bin/bash
# Round 3: the C4 line and the k = 63 variants at the last kernel commit (compact k-mer entries)
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r03_c4_final; mkdir -p $OUT
timeout 1500 python bench.py --workload c4 --no-cpu-baseline > $OUT/bench_c4.jsonl 2> $OUT/bench_c4.err; cut -c1-200 $OUT/bench_c4.jsonl
timeout 1500 python tools/perf_variants.py --k 63 --m 25 --bases 1500000000 --reads 1000000 2>$OUT/k63.err | tee $OUT/variants_k63.jsonl | cut -c1-160
tail -2 $OUT/*.err | cut -c1-200
