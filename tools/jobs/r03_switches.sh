#!/bin/bash
# Round 3, last commit: the parity suite under every measurement switch that changes which kernels run
cd "$(dirname "$0")/../.."
for e in SSHASH_AMD_INWAVE=0 SSHASH_AMD_OVERLAP=1 SSHASH_AMD_OVERLAP=0 "SSHASH_AMD_SK_SLOTS_PER_KEY=1.5 SSHASH_AMD_SK_SLOTS_PER_KMER=1.3"; do
  echo "$e: $(env $e timeout 2000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_km_sweep.py tests/test_gpu_baseline_workloads.py tests/test_gpu_streaming.py tests/test_gpu_reference_data.py -x -q -m gpu 2>&1 | grep -v '^Extension' | tail -1)"
done
