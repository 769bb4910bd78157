#!/bin/bash
# Round 3, first measurement on the calibrated stand-in: the bench line (index statistics target vs achieved, table
# histogram, table-less paths), then a kernel trace of the same command.
# usage: bash tools/jobs/r03_first.sh <name> <bench args...>
set -u
cd "$(dirname "$0")/../.."
NAME=${1:-r03_first}; shift
OUT=gpurun_out/$NAME
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --steps 10 --warmup 2 "$@" > $OUT/bench.jsonl 2> $OUT/bench.err
tail -c 600 $OUT/bench.err
BENCH="python bench.py --no-cpu-baseline --no-extra-mixes --steps 3 --warmup 1 $*"
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $BENCH > $OUT/trace.log 2>&1
find $OUT/trace -name 't_kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
head -12 $OUT/kernel_stats.csv | cut -c1-220
python3 - $OUT <<'PY'
import json, sys
r = json.loads(open(sys.argv[1] + '/bench.jsonl').read().strip().splitlines()[-1])
print({k: r[k] for k in ('value', 'ms_per_step')}, r['roofline']['frac'], r['config']['device_bytes_per_kmer'])
print(json.dumps(r['config']['table_histogram']))
print(json.dumps(r['other_paths']))
print(json.dumps(r['other_mixes']))
print({k: v['ratio'] for k, v in r['config']['index_statistics'].items() if isinstance(v, dict)})
PY
