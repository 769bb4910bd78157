#!/bin/bash
# round 6: PMC profiles of the lines whose kernels changed this round (the two streaming lines) and of the headline, same shapes as the driver's
# run -> gpurun_out/r06_prof_<record>/ ; tools/make_traffic_json.py turns each into a record of profiles/traffic.json. `all`: every line.
cd "$(dirname "$0")/../.."
P="bash tools/jobs/r06_profile.sh"
$P r06_prof_c4_streaming_p50 --workload c4 --streaming --reads 20000000 2>&1 | tail -1 | cut -c1-300
$P r06_prof_c3_streaming_p95 --workload c3 --streaming --positive 0.95 --reads 20000000 2>&1 | tail -1 | cut -c1-300
$P r06_prof_c3 --workload c3 2>&1 | tail -1 | cut -c1-300
if [ "${1:-}" = all ]; then
  export SSHASH_BENCH_CACHE=/tmp
  for w in c2 c4; do $P r06_prof_$w --workload $w 2>&1 | tail -1 | cut -c1-300; done
  SSHASH_AMD_SKTABLE=0 SSHASH_AMD_DIRECTORY=1 $P r06_prof_c3_directory --workload c3 --queries 100000000 2>&1 | tail -1 | cut -c1-300
  SSHASH_AMD_SKTABLE=0 SSHASH_AMD_DIRECTORY=0 $P r06_prof_c3_mphf --workload c3 --queries 100000000 2>&1 | tail -1 | cut -c1-300
  $P r06_prof_c3_canonical --workload c3 --canonical 2>&1 | tail -1 | cut -c1-300
fi
