#!/bin/bash
# round 5: the run-based streaming kernel against the per-base kernel (SSHASH_AMD_STREAM_WALK=bases), same box, same reads
cd "$(dirname "$0")/../.."
out=gpurun_out/${1:-r05_stream_ab}; mkdir -p $out
( time timeout 1500 python -m pytest tests/test_gpu_streaming.py tests/test_gpu_km_sweep.py -x -q -m gpu ) 2>&1 | tail -8 | tee $out/pytest_streaming.txt
S="--steps 10 --warmup 2 --no-cpu-baseline --quiet-record --reads 20000000"
run() { # name, env, args...
  name=$1; shift; envs=$1; shift
  env $envs python bench.py $S "$@" --full-record $out/$name.json > $out/$name.jsonl 2>> $out/bench.err
  python3 - $out/$name.jsonl $name <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(r['value']/1e9,2), 'G k-mers/s', r['ms_per_step'], 'ms', r['config']['report'])
PY
}
for rounds in 1 2; do
run high_hit_runs_$rounds "A=1" --workload c3 --streaming --positive 0.95
run high_hit_bases_$rounds "SSHASH_AMD_STREAM_WALK=bases" --workload c3 --streaming --positive 0.95
done
run random_runs "A=1" --workload c3 --streaming --positive 0.0
run random_bases "SSHASH_AMD_STREAM_WALK=bases" --workload c3 --streaming --positive 0.0
run c4_runs "A=1" --workload c4 --streaming
run c4_bases "SSHASH_AMD_STREAM_WALK=bases" --workload c4 --streaming
