#!/bin/bash
# round 6, first GPU check of the round's changes: the tests they touch, then the host-buffer entry points both ways, then the streaming lines
cd "$(dirname "$0")/../.."
out=gpurun_out/${1:-r06_check1}; mkdir -p $out
export TMPDIR=/tmp SSHASH_BENCH_CACHE=/tmp
timeout 2400 python -m pytest tests/test_gpu_streaming.py tests/test_gpu_parity.py tests/test_sharded.py tests/test_capi.py tests/test_gpu_bench_harness.py -m gpu -x -q > $out/pytest_touched.txt 2>&1
tail -5 $out/pytest_touched.txt
timeout 900 python tools/bench_host_path.py > $out/host_path.txt 2>&1
SSHASH_AMD_TEST_HOOKS=host_staged_copies=1 timeout 900 python tools/bench_host_path.py > $out/host_path_staged.txt 2>&1
tail -3 $out/host_path.txt $out/host_path_staged.txt
S="--steps 10 --warmup 2 --no-cpu-baseline --quiet-record --reads 20000000 --no-line-probe"
for w in "c4 0.5" "c3 0.95"; do set -- $w
  python bench.py $S --workload $1 --streaming --positive $2 --full-record $out/stream_$1.json > $out/stream_$1.jsonl 2>> $out/bench.err
  python3 -c "
import json,sys
r=json.loads(open('$out/stream_$1.jsonl').read().strip().splitlines()[-1]); print('$1', round(r['value']/1e9,2), 'G k-mers/s', r['ms_per_step'], 'ms')"
done
