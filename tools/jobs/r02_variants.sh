#!/bin/bash
# Round 2: secondary measurements (entry-point variants, query mixes, streaming, k = 63) -> gpurun_out/r02_variants/
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r02_variants
mkdir -p $OUT
timeout 1200 python tools/perf_variants.py 2>$OUT/regular.err | tee $OUT/variants_final_regular.jsonl | cut -c1-200
timeout 1200 python tools/perf_variants.py --canonical 2>$OUT/canonical.err | tee $OUT/variants_final_canonical.jsonl | cut -c1-200
timeout 1500 python tools/perf_variants.py --k 63 --m 25 --bases 1500000000 --mean-len 160 --reads 1000000 2>$OUT/k63.err | tee $OUT/variants_final_k63.jsonl | cut -c1-200
timeout 900 python bench.py --workload c2 --k 63 --m 25 --bases 1500000000 --mean-len 160 --no-cpu-baseline --no-extra-mixes 2>$OUT/bench_k63.err | tee $OUT/bench_final_k63.jsonl | cut -c1-260
timeout 900 python bench.py --workload c2 --canonical --no-cpu-baseline --no-extra-mixes 2>$OUT/bench_canonical.err | tee $OUT/bench_final_canonical.jsonl | cut -c1-260
timeout 1200 python bench.py 2>$OUT/bench_c3.err | tee $OUT/bench_c3_final.jsonl | cut -c1-260
