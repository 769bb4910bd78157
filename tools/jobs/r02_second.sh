#!/bin/bash
# Round 2: multi-slot inline keys + first/resume/deferred passes -- parity tests, C2 bench with kernel trace, C3 bench.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/${1:-r02_second}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 2>&1 | tail -15
timeout 900 python bench.py --workload c2 --no-cpu-baseline --no-extra-mixes > $OUT/bench_c2.jsonl 2> $OUT/bench_c2.err; tail -3 $OUT/bench_c2.err | cut -c1-600; cut -c1-300 $OUT/bench_c2.jsonl
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o c2 -- python bench.py --workload c2 --no-cpu-baseline --no-extra-mixes > $OUT/prof_c2.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/prof/**/c2_kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        print(r['Name'][:110], r['Calls'], r['TotalDurationNs'], r['AverageNs'])
PY
timeout 1500 python bench.py --no-cpu-baseline > $OUT/bench_c3.jsonl 2> $OUT/bench_c3.err; tail -4 $OUT/bench_c3.err | cut -c1-600; python3 - $OUT <<'PY'
import json, sys
r = json.loads(open(sys.argv[1] + '/bench_c3.jsonl').read().strip().splitlines()[-1])
print('C3', r['value'], r['ms_per_step'], r['roofline']['frac'], r['config']['device_bytes_per_kmer'], json.dumps(r['other_mixes']))
PY
