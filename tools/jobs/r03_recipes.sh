#!/bin/bash
# which of two recipes (tools/jobs/recipes_in/*.json (put the candidates there)) comes closer to the published statistics at full size?
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03_recipes
for f in tools/jobs/recipes_in/*.json; do
  cp $f sshash_amd/recipes/human_k31.json
  python bench.py --no-cpu-baseline --no-extra-mixes --no-other-paths --no-file-query --steps 3 --warmup 1 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); s=r['config']['index_statistics']
print('$f', round(r['value']/1e9,2), {k:v['ratio'] for k,v in s.items() if isinstance(v,dict) and (abs(v['ratio']-1)>0.08 or k in ('max_bucket_size','num_strings'))})" | tee -a gpurun_out/r03_recipes/out.txt
  rm -f /tmp/sshash_amd_bench_*.sshash
done
