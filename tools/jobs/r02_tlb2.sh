#!/bin/bash
# Round 2, experiment 2: (a) 1 GiB-aligned virtual range over 1 GiB physical chunks (fragment size), (b) lanes sharing a unit
# so that a 32/64-byte unit costs ONE translation request instead of one per 16-byte load instruction.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r02_tlb2
mkdir -p $OUT
export TMPDIR=/tmp
T=tools/tlb_probe
L=134217728
{
for w in 32 64; do
  timeout 300 $T 32768 $w vmm 1024 1024 $L 5 lane
  timeout 300 $T 32768 $w vmm 32768 1024 $L 5 lane
  timeout 300 $T 32768 $w vmm 2 1024 $L 5 lane
  timeout 300 $T 32768 $w malloc 0 0 $L 5 coop
  timeout 300 $T 2048 $w malloc 0 0 $L 5 coop
  timeout 300 $T 65536 $w malloc 0 0 $L 5 coop
done
timeout 300 $T 65536 32 vmm 1024 1024 $L 5 lane
timeout 300 $T 65536 64 vmm 1024 1024 $L 5 lane
timeout 300 $T 65536 64 vmm 1024 1024 $L 5 coop
} > $OUT/tlb_probe.jsonl 2> $OUT/tlb_probe.err
cut -c1-420 $OUT/tlb_probe.jsonl
cat $OUT/tlb_probe.err | head
G1="TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum"
G3="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_MISS_sum"
G4="GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"
run_pmc() { local name=$1; shift; local ctr=$1; shift
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc -o $name -- "$@" > $OUT/pmc_$name.log 2>&1; }
i=0
for g in "$G1" "$G3" "$G4"; do
  i=$((i+1))
  run_pmc v1g_w32_g$i "$g" $T 32768 32 vmm 1024 1024 $L 2 lane
  run_pmc coop_w32_g$i "$g" $T 32768 32 malloc 0 0 $L 2 coop
  run_pmc coop_w64_g$i "$g" $T 32768 64 malloc 0 0 $L 2 coop
done
python3 - <<'PY'
import csv, glob, os, collections
rows = collections.defaultdict(dict)
for f in sorted(glob.glob('gpurun_out/r02_tlb2/pmc/**/*counter_collection.csv', recursive=True)):
    name = os.path.basename(f).replace('_counter_collection.csv', '')
    run = name.rsplit('_g', 1)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'gather' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for c, v in acc.items():
        rows[run][c] = v[-1]
names = sorted(rows)
ctrs = sorted({c for r in rows.values() for c in r})
print('%-45s' % 'counter', *['%14s' % n for n in names])
for c in ctrs:
    print('%-45s' % c, *['%14.4g' % rows[n].get(c, float('nan')) for n in names])
PY
