#!/bin/bash
# Round 2, experiment 1: address translation vs DRAM as the cause of the large-array random-read ceiling.
# Usage on the GPU box (through gpurun): bash tools/jobs/r02_tlb.sh
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r02_tlb
mkdir -p $OUT
export TMPDIR=/tmp
T=tools/tlb_probe
{
for mib in 2048 8192 32768 65536; do
  for w in 8 16 32 64; do timeout 120 $T $mib $w malloc; done
done
# virtual-memory API: (chunk, alignment) sweeps the largest page-table fragment the driver can form
for w in 32 64; do
  timeout 300 $T 32768 $w vmm 2 2
  timeout 300 $T 32768 $w vmm 64 64
  timeout 300 $T 32768 $w vmm 1024 1024
  timeout 300 $T 32768 $w vmm 1024 2
  timeout 300 $T 32768 $w vmm 32768 1024
done
timeout 300 $T 65536 32 vmm 1024 1024
timeout 300 $T 65536 32 vmm 65536 1024
timeout 300 $T 32768 32 pieces 1024
timeout 300 $T 32768 32 pieces 64
} > $OUT/tlb_probe.jsonl 2> $OUT/tlb_probe.err
cat $OUT/tlb_probe.jsonl | cut -c1-400

# counters: one group per pass (no trace domains together with --pmc)
G1="TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum"
G2="TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum"
G3="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_MISS_sum"
G4="GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"
G5="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum"
G6="TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_REQ_sum TCC_HIT_sum"
run_pmc() {  # name, counters, command...
  local name=$1; shift; local ctr=$1; shift
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc -o $name -- "$@" > $OUT/pmc_$name.log 2>&1
}
i=0
for g in "$G1" "$G2" "$G3" "$G4" "$G5" "$G6"; do
  i=$((i+1))
  run_pmc m2g_w32_g$i  "$g" $T 2048 32 malloc 0 0 134217728 2
  run_pmc m32g_w32_g$i "$g" $T 32768 32 malloc 0 0 134217728 2
  run_pmc m32g_w64_g$i "$g" $T 32768 64 malloc 0 0 134217728 2
  run_pmc m32g_w8_g$i  "$g" $T 32768 8 malloc 0 0 134217728 2
  run_pmc v32g_w32_g$i "$g" $T 32768 32 vmm 1024 1024 134217728 2
done
find $OUT/pmc -name '*counter_collection.csv' | head -50
python3 - <<'PY'
import csv, glob, os, collections
rows = collections.defaultdict(dict)
for f in sorted(glob.glob('gpurun_out/r02_tlb/pmc/**/*counter_collection.csv', recursive=True)):
    name = os.path.basename(f).replace('_counter_collection.csv', '')
    run = name.rsplit('_g', 1)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'gather' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for c, v in acc.items():
        rows[run][c] = v[-1]  # last launch
for run, d in rows.items():
    print(run, {k: int(v) for k, v in sorted(d.items())})
PY
