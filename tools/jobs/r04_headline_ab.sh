#!/bin/bash
# same-box A/B of the headline (C3) and of C4: the library at commit 1f5ef2d (tools/debug/libsshash_amd_old.so) against the one in the tree
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_headline_ab; mkdir -p $out
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-mixes --no-file-query --no-other-paths --no-other-workloads "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']/1e9,2), r['ms_per_step'], r['roofline']['avg_kernel_ms'] if 'avg_kernel_ms' in r['roofline'] else '')"; }
{
for wl in c3 c4; do
  for round in 1 2; do
    echo -n "$wl old: "; SSHASH_AMD_LIBRARY=$PWD/tools/debug/libsshash_amd_old.so run --workload $wl
    echo -n "$wl new: "; run --workload $wl
  done
done
} 2>&1 | tee $out/ab.txt
bash tools/jobs/r04_ifetch.sh 2>&1 | tail -24
