#!/bin/bash
# Same-box comparison of several builds of the library (tools/debug/libsshash_amd_<tag>.so, plus "cur" = the one in place).
# usage: TAGS="old sc1_nt ..." ROUNDS=4 r02_ab_multi.sh <bench args>
set -u
cd "$(dirname "$0")/../.."
B="python bench.py --no-cpu-baseline --no-extra-mixes $*"
cp sshash_amd/libsshash_amd.so /tmp/cur.so
for round in $(seq 1 ${ROUNDS:-4}); do
  for tag in cur $TAGS; do
    if [ $tag = cur ]; then cp /tmp/cur.so sshash_amd/libsshash_amd.so; else cp tools/debug/libsshash_amd_$tag.so sshash_amd/libsshash_amd.so; fi
    echo -n "$tag "; $B 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']/1e9,2))"
  done
done | python -c "
import sys,collections
d=collections.defaultdict(list)
for l in sys.stdin:
    k,v=l.split(); d[k].append(float(v))
for k,v in d.items(): print(k.ljust(14), sorted(v), round(sum(v)/len(v),2))"
cp /tmp/cur.so sshash_amd/libsshash_amd.so
