#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03_k63b
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r03_k63b/pytest.log 2>&1
tail -3 gpurun_out/r03_k63b/pytest.log
for i in 1 2 3; do
python bench.py --workload c2 --k 63 --m 25 --bases 1500000000 --no-cpu-baseline --no-extra-mixes --no-other-paths --no-file-query 2>>gpurun_out/r03_k63b/bench.err | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); print('k63', r['value']/1e9, r['ms_per_step'])"
done | tee gpurun_out/r03_k63b/k63.txt
python bench.py --no-cpu-baseline --no-extra-mixes --no-other-paths --no-file-query 2>>gpurun_out/r03_k63b/bench.err | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); print('c3', r['value']/1e9, r['ms_per_step'])"
