#!/bin/bash
# k = 63 with the 31-base table key (default now): slots per item in the keys' region once more, same box, alternating
cd "$(dirname "$0")/../.."
out=gpurun_out/${NAME:-r04_c4_slots2}; mkdir -p $out
B="python bench.py --no-cpu-baseline --no-extra-mixes --no-file-query --no-other-workloads --no-other-paths --no-line-probe --steps 10 --warmup 2 --workload c4"
val() { python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']/1e9,2), r['ms_per_step'], r['config']['device_bytes_per_kmer'])"; }
( for round in 1 2 3 4; do
    for s in 2.5 3.0 3.5; do
      echo -n "c4 (table key 31) $s slots per key: "; SSHASH_AMD_SK_SLOTS_PER_KEY=$s $B 2>/dev/null | val
    done
  done ) 2>&1 | tee $out/slots_per_key_c4_table_key_31.txt
