#!/bin/bash
# Same-box A/B of two builds over the entry-point variants (tools/perf_variants.py, lookups only). Prints variant, old, new rates (G/s), two rounds.
set -u
cd "$(dirname "$0")/../.."
cp sshash_amd/libsshash_amd.so /tmp/new.so
for round in 1 2; do
  for which in old new; do
    if [ $which = old ]; then cp tools/debug/libsshash_amd_old.so sshash_amd/libsshash_amd.so; else cp /tmp/new.so sshash_amd/libsshash_amd.so; fi
    python tools/perf_variants.py --skip-streaming $* 2>/dev/null > /tmp/v_$which.jsonl
  done
  python3 - <<'PY'
import json
o={json.loads(l)['variant']:json.loads(l)['rate'] for l in open('/tmp/v_old.jsonl')}
n={json.loads(l)['variant']:json.loads(l)['rate'] for l in open('/tmp/v_new.jsonl')}
for k in o:
    if 'host' in k: continue
    print(k.ljust(40), round(o[k]/1e9,2), round(n[k]/1e9,2), '%+.1f%%' % (100*(n[k]/o[k]-1)))
PY
done
cp /tmp/new.so sshash_amd/libsshash_amd.so
