#!/bin/bash
# instruction fetch of the streaming kernel (k = 31, high-hit set): is a step bound by the instruction cache?
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r04_ifetch; mkdir -p $OUT; export TMPDIR=/tmp
BENCH="python bench.py --no-cpu-baseline --steps 3 --warmup 1 --workload c3 --bases 1000000000 --streaming --reads 20000000 --positive 0.95 --stream-oracle-reads 20000"
$BENCH > $OUT/bench.jsonl 2> $OUT/bench.err
i=0
for g in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_BRANCH SQ_WAVES" \
         "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES SQC_TC_INST_REQ" \
         "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_BUSY_CU_CYCLES SQ_CYCLES"; do
  i=$((i+1))
  timeout 1200 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $OUT/pmc -o g$i -- $BENCH > $OUT/pmc_g$i.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, sys, json, collections
out = sys.argv[1]
res = collections.defaultdict(list)
for f in sorted(glob.glob(out + '/pmc/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if 'streaming_kernel' in r['Kernel_Name']: res[r['Counter_Name']].append(float(r['Counter_Value']))
summary = {c: sum(x for x in v if x >= 0.5 * max(v)) / max(1, len([x for x in v if x >= 0.5 * max(v)])) for c, v in res.items()}
json.dump(summary, open(out + '/summary.json', 'w'), indent=1)
print(json.dumps(summary, indent=1))
PY
rm -rf $OUT/pmc
