#!/bin/bash
# Round 3: the deferred pass on the auxiliary stream, now that it is the only tail pass (k <= 31) -- same-box A/B
cd "$(dirname "$0")/../.."
bash tools/jobs/r03_env_ab.sh r03_overlap2 SSHASH_AMD_OVERLAP 0 1 3
bash tools/jobs/r03_env_ab.sh r03_overlap2 SSHASH_AMD_OVERLAP 0 1 2 --workload c2
