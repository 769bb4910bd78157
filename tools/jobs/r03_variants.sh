#!/bin/bash
# Round 3: secondary measurements at the round's last kernel change (entry-point variants, query mixes, streaming, canonical, k = 63, C2 line)
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r03_variants
mkdir -p $OUT
timeout 1200 python tools/perf_variants.py 2>$OUT/regular.err | tee $OUT/variants_regular.jsonl | cut -c1-220
timeout 1200 python tools/perf_variants.py --canonical 2>$OUT/canonical.err | tee $OUT/variants_canonical.jsonl | cut -c1-220
timeout 1500 python tools/perf_variants.py --k 63 --m 25 --bases 1500000000 --reads 1000000 2>$OUT/k63.err | tee $OUT/variants_k63.jsonl | cut -c1-220
timeout 900 python bench.py --workload c2 --k 63 --m 25 --bases 1500000000 --no-cpu-baseline --no-extra-mixes --no-other-paths --no-file-query 2>$OUT/bench_k63.err | tee $OUT/bench_k63.jsonl | cut -c1-260
timeout 900 python bench.py --workload c2 --canonical --no-cpu-baseline --no-extra-mixes --no-file-query 2>$OUT/bench_canonical.err | tee $OUT/bench_c2_canonical.jsonl | cut -c1-260
timeout 900 python bench.py --workload c2 2>$OUT/bench_c2.err | tee $OUT/bench_c2.jsonl | cut -c1-260
tail -3 $OUT/*.err | cut -c1-300
