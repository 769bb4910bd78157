#!/bin/bash
# Round 2: instruction counters only (one PMC pass) of a bench command. usage: r02_sq_only.sh <name> <bench args...>
set -u
cd "$(dirname "$0")/../.."
NAME=$1; shift
OUT=gpurun_out/$NAME
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python bench.py --no-cpu-baseline --no-extra-mixes --steps 3 --warmup 1 $*"
$BENCH > $OUT/bench.jsonl 2> $OUT/bench.err
timeout 1200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc -o g -- $BENCH > $OUT/pmc.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + '/pmc/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'lookup_kernel' not in n: continue
        key = 'fast' if 'fast_lookup' in n else 'resume' if 'resume_lookup' in n else 'deferred'
        res[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in res.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    w = m.get('SQ_WAVES', 1)
    print(k, 'waves', int(w), 'VALU/wave', round(m.get('SQ_INSTS_VALU', 0) / w, 1), 'SALU/wave', round(m.get('SQ_INSTS_SALU', 0) / w, 1),
          'LDS/wave', round(m.get('SQ_INSTS_LDS', 0) / w, 1), 'VMEM_RD/wave', round(m.get('SQ_INSTS_VMEM_RD', 0) / w, 1),
          'wave quad-cycles', round(m.get('SQ_WAVE_CYCLES', 0) / w), 'GUI_ACTIVE', int(m.get('GRBM_GUI_ACTIVE', 0)))
PY
tail -1 $OUT/bench.jsonl | cut -c1-170
