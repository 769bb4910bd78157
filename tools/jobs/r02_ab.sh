#!/bin/bash
# Same-box A/B of two builds of the library: tools/debug/libsshash_amd_old.so against sshash_amd/libsshash_amd.so.
# usage: r02_ab.sh <bench args...>   (box-to-box spread of one command is +-5 %: only a same-box comparison tells)
set -u
cd "$(dirname "$0")/../.."
B="python bench.py --no-cpu-baseline --no-extra-mixes $*"
cp sshash_amd/libsshash_amd.so /tmp/new.so
for round in $(seq 1 ${ROUNDS:-3}); do
  cp tools/debug/libsshash_amd_old.so sshash_amd/libsshash_amd.so; echo -n "old: "; $B 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']/1e9,2), r['ms_per_step'])"
  cp /tmp/new.so sshash_amd/libsshash_amd.so;                      echo -n "new: "; $B 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']/1e9,2), r['ms_per_step'])"
done
