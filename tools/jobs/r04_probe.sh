#!/bin/bash
# the bench harness tests (the default run now ends with the random-line probe of its box), then the driver's command
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_probe; mkdir -p $out
timeout 2400 python -m pytest tests/test_gpu_bench_harness.py -x -q -m gpu 2>&1 | tail -4 | tee $out/pytest.txt
bash tools/jobs/r04_driver_command.sh 2>&1 | tail -14 | tee $out/driver_command.txt
python3 - <<'PY' | tee -a $out/driver_command.txt
import json
r=json.loads(open('gpurun_out/r04_driver_command/bench.jsonl').read().strip().splitlines()[-1])
print('c3', r['roofline']['random_unit_bound'])
for k in ('c2','c4'): print(k, r['other_workloads'][k]['roofline']['random_unit_bound'])
PY
