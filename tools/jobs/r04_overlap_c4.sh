#!/bin/bash
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_overlap_c4; mkdir -p $out
B="python bench.py --no-cpu-baseline --no-extra-mixes --no-file-query --no-other-paths --no-other-workloads --steps 10 --warmup 2 --workload c4"
val() { python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']/1e9,2), r['ms_per_step'])"; }
( for round in 1 2 3; do
    echo -n "c4 tail passes behind the first pass: "; $B 2>/dev/null | val
    echo -n "c4 tail passes on the auxiliary stream (SSHASH_AMD_OVERLAP=1): "; SSHASH_AMD_OVERLAP=1 $B 2>/dev/null | val
  done ) 2>&1 | tee $out/ab.txt
