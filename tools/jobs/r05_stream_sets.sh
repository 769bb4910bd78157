#!/bin/bash
# round 5: the streaming query on the three read sets of RESULTS.md (k = 31 high-hit, k = 31 random reads, config C4's), one box
cd "$(dirname "$0")/../.."
out=gpurun_out/${1:-r05_stream_sets}; mkdir -p $out
S="--steps 10 --warmup 2 --no-cpu-baseline --quiet-record --reads 20000000"
run() { # name, args...
  name=$1; shift
  python bench.py $S "$@" --full-record $out/$name.json > $out/$name.jsonl 2>> $out/bench.err
  python3 - $out/$name.jsonl $name <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(r['value']/1e9,2), 'G k-mers/s', r['ms_per_step'], 'ms')
PY
}
run high_hit --workload c3 --streaming --positive 0.95
run random --workload c3 --streaming --positive 0.0
run c4 --workload c4 --streaming
