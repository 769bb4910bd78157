#!/bin/bash
# the opt-in test of a dictionary with more than 2^32 k-mer starts (4.6 G bases; the generator's global de-duplication at that size)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r05_huge
( time SSHASH_TEST_HUGE=1 timeout 2400 python -m pytest tests/test_gpu_baseline_workloads.py -x -q -m gpu -k "2_32" ) 2>&1 | tail -15 | tee gpurun_out/r05_huge/pytest.txt
