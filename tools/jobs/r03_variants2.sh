#!/bin/bash
# Round 3: the entry-point variants again with the streaming reads drawn from the stand-in the index was built from
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r03_variants2
mkdir -p $OUT
timeout 1200 python tools/perf_variants.py 2>$OUT/regular.err | tee $OUT/variants_regular.jsonl | cut -c1-260
timeout 1200 python tools/perf_variants.py --canonical 2>$OUT/canonical.err | tee $OUT/variants_canonical.jsonl | cut -c1-260
timeout 1500 python tools/perf_variants.py --k 63 --m 25 --bases 1500000000 --reads 1000000 2>$OUT/k63.err | tee $OUT/variants_k63.jsonl | cut -c1-260
tail -3 $OUT/*.err | cut -c1-300
