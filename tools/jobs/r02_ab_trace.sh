#!/bin/bash
# Same-box kernel times of two builds (tools/debug/libsshash_amd_old.so against sshash_amd/libsshash_amd.so). usage: r02_ab_trace.sh <bench args>
set -u
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-extra-mixes --steps 3 --warmup 1 $*"
cp sshash_amd/libsshash_amd.so /tmp/new.so
for which in old new; do
  if [ $which = old ]; then cp tools/debug/libsshash_amd_old.so sshash_amd/libsshash_amd.so; else cp /tmp/new.so sshash_amd/libsshash_amd.so; fi
  rm -rf /tmp/tr_$which
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$which -o t -- $B > /tmp/tr_$which.log 2>&1
  f=$(find /tmp/tr_$which -name 't_kernel_stats.csv' | head -1)
  echo "== $which"; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'lookup_kernel' in r['Name']:
        print(r['Name'][17:60], r['Calls'], round(float(r['AverageNs'])/1e6, 3))
PY
done
cp /tmp/new.so sshash_amd/libsshash_amd.so
