#!/bin/bash
# Round 2: kernel times of the streaming paths (counters kernel; encode -> masked lookup -> classify)
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r02_stream
mkdir -p $OUT
export TMPDIR=/tmp
B="python tools/perf_variants.py --queries 1000000 $*"
$B > $OUT/variants.jsonl 2> $OUT/variants.err
grep streaming $OUT/variants.jsonl | cut -c1-220
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $B > $OUT/trace.log 2>&1
f=$(find $OUT/trace -name 't_kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:22]:
    print(r['Name'][:110].ljust(110), r['Calls'], round(float(r['AverageNs'])/1e6,3), r['Percentage'])
PY
