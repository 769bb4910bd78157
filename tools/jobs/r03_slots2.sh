#!/bin/bash
cd "$(dirname "$0")/../.."
bash tools/jobs/r03_env_ab.sh r03_slots2 SSHASH_AMD_SK_SLOTS_PER_KEY 1.75 1.5 2
bash tools/jobs/r03_env_ab.sh r03_slots2 SSHASH_AMD_SK_SLOTS_PER_KEY 1.25 2.25 2
grep -h "replica in HBM" gpurun_out/r03_slots2/bench.err | cut -c1-400 | sed 's/directory_sectors.*sk_keys/ .. sk_keys/'
