#!/bin/bash
# round 6: the driver's flow (index built in-process the first time, 20 steps) with round 5's library against this round's, alternating, one box
cd "$(dirname "$0")/../.."
out=gpurun_out/${1:-r06_lib_ab}; mkdir -p $out
export SSHASH_BENCH_CACHE=/tmp
S="--steps 20 --warmup 5 --no-other-workloads --no-file-query --no-other-paths --quiet-record"
for round in 1 2 3; do for lib in "" tools/ab/lib_round5.so; do
  tag=${round}_$(basename ${lib:-round6} .so)
  SSHASH_AMD_LIBRARY=${lib:+$PWD/$lib} python bench.py $S --full-record $out/$tag.json > $out/$tag.jsonl 2>> $out/bench.err
  python3 -c "
import json; r=json.load(open('$out/$tag.json')); print('$tag', round(r['value']/1e9,2), 'G lookups/s', r['ms_per_step'], 'ms/step', {k:round(v['lookups_per_s']/1e9,2) for k,v in r['other_mixes'].items()}, 'probe', r['roofline']['random_unit_bound'].get('this_box',{}).get('every_allocation_G_per_s'))" | tee -a $out/runs.txt
done; done
