#!/bin/bash
# the driver's command, as the driver runs it
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_driver_command; mkdir -p $out
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.jsonl 2> $out/bench.err ) 2>&1 | tail -3
tail -c 600 $out/bench.err; python3 - <<'PY'
import json
r=json.loads(open('gpurun_out/r04_driver_command/bench.jsonl').read().strip().splitlines()[-1])
print('c3', r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline'].get('frac_hbm_traffic'), r['config']['device_bytes_per_kmer'])
print('other_paths', {k:(v['lookups_per_s'], v['roofline_frac']) for k,v in (r['other_paths'] or {}).items()})
print('file', {k: v.get('kmers_per_s') for k,v in r['streaming_from_file'].items() if isinstance(v, dict) and 'kmers_per_s' in v})
for k,v in (r['other_workloads'] or {}).items():
    print(k, v.get('value'), v.get('unit'), v.get('ms_per_step'), (v.get('roofline') or {}).get('frac'), (v.get('roofline') or {}).get('frac_hbm_traffic'), v.get('wall_s_of_the_child'), v.get('error'))
PY
