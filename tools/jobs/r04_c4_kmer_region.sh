#!/bin/bash
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_c4_kmer_region; mkdir -p $out
B="python bench.py --no-cpu-baseline --no-extra-mixes --no-file-query --no-other-paths --no-other-workloads --steps 10 --warmup 2 --workload c4"
val() { python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']/1e9,2), r['ms_per_step'], r['config']['device_bytes_per_kmer'], r['config']['device_stats']['sk_deferred_keys'])"; }
( for round in 1 2 3; do for skm in 1.75 2.0 2.5; do
    echo -n "c4 places per heavy k-mer $skm: "; SSHASH_AMD_SK_SLOTS_PER_KMER=$skm $B 2>/dev/null | val
  done; done ) 2>&1 | tee $out/ab.txt
