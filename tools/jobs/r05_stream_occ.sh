#!/bin/bash
# round 5: the run-based streaming kernel compiled for 4 / 5 / 6 waves a SIMD (SSHASH_AMD_STREAM_OCC), same box, same reads
cd "$(dirname "$0")/../.."
out=gpurun_out/${1:-r05_stream_occ}; mkdir -p $out
( time timeout 1500 python -m pytest tests/test_gpu_streaming.py tests/test_gpu_km_sweep.py -x -q -m gpu ) 2>&1 | tail -8 | tee $out/pytest_streaming.txt
S="--steps 10 --warmup 2 --no-cpu-baseline --quiet-record --reads 20000000"
run() { # name, env, args...
  name=$1; shift; envs=$1; shift
  env $envs python bench.py $S "$@" --full-record $out/$name.json > $out/$name.jsonl 2>> $out/bench.err
  python3 - $out/$name.jsonl $name <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(r['value']/1e9,2), 'G k-mers/s', r['ms_per_step'], 'ms')
PY
}
for occ in ${OCCS:-4 5 6}; do
run high_hit_occ$occ "SSHASH_AMD_STREAM_OCC=$occ" --workload c3 --streaming --positive 0.95
run random_occ$occ "SSHASH_AMD_STREAM_OCC=$occ" --workload c3 --streaming --positive 0.0
run c4_occ$occ "SSHASH_AMD_STREAM_OCC=$occ" --workload c4 --streaming
done
