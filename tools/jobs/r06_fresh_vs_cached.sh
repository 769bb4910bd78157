#!/bin/bash
# round 6: the driver's command gave C3 27.0 ms/step in the process that BUILT the index and 24.4 in later processes of the same box that loaded it
# from the cache. Which is it -- building in-process, or being the first process on the box?
cd "$(dirname "$0")/../.."
out=gpurun_out/${1:-r06_fresh_vs_cached}; mkdir -p $out
S="--steps 20 --warmup 5 --no-cpu-baseline --no-extra-mixes --no-file-query --no-other-paths --no-other-workloads --no-line-probe --quiet-record"
run() { # name, cache dir
  SSHASH_BENCH_CACHE=$2 python bench.py $S --full-record $out/$1.json > $out/$1.jsonl 2>> $out/bench.err
  python3 -c "
import json; r=json.load(open('$out/$1.json')); print('$1', round(r['value']/1e9,2), 'G lookups/s', r['ms_per_step'], 'ms/step; upload', r['per_rank'][0]['upload_s'], 's')" | tee -a $out/runs.txt
}
mkdir -p /tmp/cx /tmp/cy
run A_builds_index /tmp/cx
run B_loads_cache /tmp/cx
run C_builds_index_again /tmp/cy
run D_loads_cache /tmp/cy
run E_loads_cache /tmp/cx
