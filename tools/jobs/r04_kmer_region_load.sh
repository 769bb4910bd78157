#!/bin/bash
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_kmer_region_load; mkdir -p $out
B="python bench.py --no-cpu-baseline --no-extra-mixes --no-file-query --no-other-paths --no-other-workloads --steps 10 --warmup 2"
val() { python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']/1e9,2), r['ms_per_step'], r['config']['device_bytes_per_kmer'], r['config']['device_stats']['sk_deferred_keys'])"; }
( for round in 1 2 3; do for skm in 2.0 1.75 1.5; do
    echo -n "c3 places per heavy k-mer $skm: "; SSHASH_AMD_SK_SLOTS_PER_KMER=$skm $B --workload c3 2>/dev/null | val
  done; done
  for round in 1 2; do for skm in 2.0 1.5; do
    echo -n "c2 places per heavy k-mer $skm: "; SSHASH_AMD_SK_SLOTS_PER_KMER=$skm $B --workload c2 2>/dev/null | val
  done; done ) 2>&1 | tee $out/ab.txt
