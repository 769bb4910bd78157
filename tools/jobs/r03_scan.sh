#!/bin/bash
# table-less paths after the wave-cooperative bucket scan: parity, then the bench's other_paths + kernel trace
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r03_scan
mkdir -p $OUT
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_sharded.py tests/test_gpu_km_sweep.py -x -q -m gpu > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
export TMPDIR=/tmp
BENCH="python bench.py --no-cpu-baseline --no-extra-mixes --steps 5 --warmup 2"
$BENCH > $OUT/bench.jsonl 2> $OUT/bench.err
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $BENCH > $OUT/trace.log 2>&1
find $OUT/trace -name 't_kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
python3 - $OUT <<'PY'
import json, sys, csv
r = json.loads(open(sys.argv[1] + '/bench.jsonl').read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step'], json.dumps(r['other_paths']))
for row in csv.DictReader(open(sys.argv[1] + '/kernel_stats.csv')):
    if 'lookup_kernel' in row['Name']:
        print(row['Name'][:70], row['Calls'], round(float(row['AverageNs']) / 1e6, 3))
PY
