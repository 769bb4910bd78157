#!/bin/bash
# k = 31 streaming query on high-hit reads against the table's key length (longer keys: fewer seeds under heavy keys, more items): same box, alternating
cd "$(dirname "$0")/../.."
out=gpurun_out/${NAME:-r04_stream_key}; mkdir -p $out
sval() { python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']/1e9,2), r['ms_per_step'])"; }
( for round in 1 2; do
    for m in 21 23 25; do echo -n "c3 streaming high-hit, table key $m: "; SSHASH_AMD_SK_M=$m python bench.py --streaming --reads 20000000 --steps 5 --warmup 1 --stream-oracle-reads 20000 --workload c3 --positive 0.95 2>/dev/null | sval; done
  done ) 2>&1 | tee $out/streaming_k31_table_key_length.txt
