#!/bin/bash
# k = 63 streaming query, previous build against this one (compact k-mer entries), same box
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03_stream_ab
B="python tools/perf_variants.py --k 63 --m 25 --bases 1500000000 --reads 1000000"
$B > /dev/null 2>&1
cp sshash_amd/libsshash_amd.so /tmp/new.so
for round in 1 2; do
  cp tools/debug/libsshash_amd_old.so sshash_amd/libsshash_amd.so; $B 2>/dev/null | grep streaming | python -c "
import sys,json
for l in sys.stdin: r=json.loads(l); print('old', r['variant'], round(r['rate']/1e9,2))"
  cp /tmp/new.so sshash_amd/libsshash_amd.so; $B 2>/dev/null | grep streaming | python -c "
import sys,json
for l in sys.stdin: r=json.loads(l); print('new', r['variant'], round(r['rate']/1e9,2))"
done | tee gpurun_out/r03_stream_ab/ab.txt
cp /tmp/new.so sshash_amd/libsshash_amd.so
