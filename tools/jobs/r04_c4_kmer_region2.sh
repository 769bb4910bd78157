#!/bin/bash
# k = 63 at the round's last defaults (31-base key, 3.0 slots per item): places per k-mer in the heavy keys' k-mers' region, same box, alternating
cd "$(dirname "$0")/../.."
out=gpurun_out/${NAME:-r04_c4_kmer_region2}; mkdir -p $out
B="python bench.py --no-cpu-baseline --no-extra-mixes --no-file-query --no-other-workloads --no-other-paths --no-line-probe --steps 10 --warmup 2 --workload c4"
val() { python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']/1e9,2), r['ms_per_step'], r['config']['device_bytes_per_kmer'])"; }
sval() { python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']/1e9,2), r['ms_per_step'])"; }
( for round in 1 2 3; do
    for s in 1.75 2.5; do echo -n "c4 $s places per heavy k-mer: "; SSHASH_AMD_SK_SLOTS_PER_KMER=$s $B 2>/dev/null | val; done
  done
  for round in 1 2; do
    for s in 1.75 2.5; do echo -n "c4 streaming, $s places per heavy k-mer: "; SSHASH_AMD_SK_SLOTS_PER_KMER=$s python bench.py --streaming --reads 20000000 --steps 5 --warmup 1 --stream-oracle-reads 20000 --workload c4 2>/dev/null | sval; done
  done ) 2>&1 | tee $out/kmer_region_places_c4_last_defaults.txt
