#!/bin/bash
cd "$(dirname "$0")/../.."
export SSHASH_AMD_VERBOSE=1
bash tools/jobs/r03_env_ab.sh r03_alloc SSHASH_AMD_SK_SLOTS_PER_KMER 1.75 3.5 2 --workload c4
bash tools/jobs/r03_env_ab.sh r03_alloc SSHASH_AMD_SK_SLOTS_PER_KMER 2.6 1.3 1 --workload c4
grep -h "super-k-mer table" gpurun_out/r03_alloc/bench.err
