#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03_c4
timeout 1500 python -m pytest tests/test_gpu_baseline_workloads.py tests/test_gpu_bench_harness.py -x -q -m gpu > gpurun_out/r03_c4/pytest.log 2>&1; tail -3 gpurun_out/r03_c4/pytest.log
timeout 1500 python bench.py --workload c4 --no-cpu-baseline --no-other-paths --steps 5 --warmup 2 > gpurun_out/r03_c4/bench_c4.jsonl 2> gpurun_out/r03_c4/bench_c4.err
tail -4 gpurun_out/r03_c4/bench_c4.err | cut -c1-400
python3 - <<'PY'
import json
r=json.loads(open('gpurun_out/r03_c4/bench_c4.jsonl').read().strip().splitlines()[-1])
print(r['value']/1e9, r['ms_per_step'], r['roofline']['frac'], r['config']['device_bytes_per_kmer'], r['config']['device_index_bytes'])
print({k:v['ratio'] for k,v in r['config']['index_statistics'].items() if isinstance(v,dict)})
print(r['config']['table_histogram'])
print(r['other_mixes'])
PY
