#!/bin/bash
# Round 3: the heavy keys' k-mers in a region of their own -- parity, then its load factor on C3 and C4 (same box)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03_kmer_region
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_workloads.py tests/test_gpu_streaming.py -x -q -m gpu 2>&1 | tail -3
SSHASH_AMD_SK_SLOTS_PER_KMER=1.5 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_km_sweep.py -x -q -m gpu 2>&1 | tail -3
bash tools/jobs/r03_env_ab.sh r03_kmer_region SSHASH_AMD_SK_SLOTS_PER_KMER 2.5 1.75 2
bash tools/jobs/r03_env_ab.sh r03_kmer_region SSHASH_AMD_SK_SLOTS_PER_KMER 1.5 2.0 1
bash tools/jobs/r03_env_ab.sh r03_kmer_region SSHASH_AMD_SK_SLOTS_PER_KMER 2.5 1.75 2 --workload c4
bash tools/jobs/r03_env_ab.sh r03_kmer_region SSHASH_AMD_SK_SLOTS_PER_KMER 1.5 2.0 1 --workload c4
grep -h "replica in HBM" gpurun_out/r03_kmer_region/bench.err | cut -c1-420
