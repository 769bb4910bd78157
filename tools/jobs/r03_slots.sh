#!/bin/bash
cd "$(dirname "$0")/../.."
bash tools/jobs/r03_env_ab.sh r03_slots SSHASH_AMD_SK_SLOTS_PER_KEY 2.5 2.0 2
bash tools/jobs/r03_env_ab.sh r03_slots SSHASH_AMD_SK_SLOTS_PER_KEY 1.75 3.0 2
grep -h "replica in HBM" gpurun_out/r03_slots/bench.err | cut -c1-60
