#!/bin/bash
# round 5, last: PMC records of the streaming lines at the final kernel, the host-buffer entry point, the whole GPU suite + smoke
cd "$(dirname "$0")/../.."
P="bash tools/jobs/r05_profile.sh"
$P r05_prof_c3_streaming_p95 --workload c3 --streaming --positive 0.95 --reads 20000000 2>&1 | tail -1 | cut -c1-200
$P r05_prof_c4_streaming_p50 --workload c4 --streaming --reads 20000000 2>&1 | tail -1 | cut -c1-200
mkdir -p gpurun_out/r05_host_path; python tools/bench_host_path.py 2>&1 | grep -v amdgpu.ids | tail -3 | tee gpurun_out/r05_host_path/out.txt
bash tools/jobs/r05_suite.sh r05_suite_final 2>&1 | tail -12
