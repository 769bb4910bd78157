#!/bin/bash
# compact k-mer entries (k <= 63): old build / new build / new build with the region padded to the old size, alternating, same box
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03_compact_ab2
B="python bench.py --workload c4 --no-cpu-baseline --no-extra-mixes --no-file-query --no-other-paths --steps 10 --warmup 2"
$B > /dev/null 2>&1
cp sshash_amd/libsshash_amd.so /tmp/new.so
P='import json,sys; r=json.loads(sys.stdin.read()); print(round(r["value"]/1e9,2), r["ms_per_step"], r["config"]["device_index_bytes"])'
for round in 1 2 3 4 5; do
  cp tools/debug/libsshash_amd_old.so sshash_amd/libsshash_amd.so; echo -n "old: "; $B 2>/dev/null | python -c "$P"
  cp /tmp/new.so sshash_amd/libsshash_amd.so; echo -n "new: "; $B 2>/dev/null | python -c "$P"
  echo -n "new_padded: "; SSHASH_AMD_SK_SLOTS_PER_KMER=3.5 $B 2>/dev/null | python -c "$P"
done | tee gpurun_out/r03_compact_ab2/ab.txt
cp /tmp/new.so sshash_amd/libsshash_amd.so
