#!/bin/bash
# round 5: PMC profiles of the streaming lines of the driver's run (same shapes: 20 M reads) -> gpurun_out/r05_prof_*/summary.json (tools/make_traffic_json.py)
cd "$(dirname "$0")/../.."
bash tools/jobs/r05_profile.sh r05_prof_c3_streaming_p95 --workload c3 --streaming --positive 0.95 --reads 20000000 2>&1 | tail -6 | cut -c1-1500
bash tools/jobs/r05_profile.sh r05_prof_c4_streaming_p50 --workload c4 --streaming --reads 20000000 2>&1 | tail -6 | cut -c1-1500
bash tools/jobs/r05_profile.sh r05_prof_c3_streaming_p0 --workload c3 --streaming --positive 0.0 --reads 20000000 2>&1 | tail -6 | cut -c1-1500
( time timeout 1500 python -m pytest tests/test_gpu_baseline_workloads.py -x -q -m gpu -k "full_size" ) 2>&1 | tail -8 | tee gpurun_out/r05_prof_c3_streaming_p95/pytest_full_size.txt
