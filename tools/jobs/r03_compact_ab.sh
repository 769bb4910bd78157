#!/bin/bash
# Round 3: compact k-mer entries (k <= 63), same-box A/B against the previous build (tools/debug/libsshash_amd_old.so), then the k = 63 parity files
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03_compact_ab
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r03_compact_ab/pytest.log 2>&1; grep -v "^Extension" gpurun_out/r03_compact_ab/pytest.log | tail -3
ROUNDS=3 bash tools/jobs/r02_ab.sh --workload c4 --no-file-query --no-other-paths --steps 10 --warmup 2 | tee gpurun_out/r03_compact_ab/ab_c4.txt
