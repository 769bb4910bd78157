#!/bin/bash
# round 5: PMC profiles of every line of the driver's run (same shapes) -> gpurun_out/r05_prof_<record>/ ; tools/make_traffic_json.py turns each into a record of profiles/traffic.json
cd "$(dirname "$0")/../.."
P="bash tools/jobs/r05_profile.sh"
for w in ${WORKLOADS:-c3 c2 c4}; do $P r05_prof_$w --workload $w 2>&1 | tail -1 | cut -c1-300; done
$P r05_prof_c3_streaming_p95 --workload c3 --streaming --positive 0.95 --reads 20000000 2>&1 | tail -1 | cut -c1-300
$P r05_prof_c4_streaming_p50 --workload c4 --streaming --reads 20000000 2>&1 | tail -1 | cut -c1-300
SSHASH_AMD_SKTABLE=0 SSHASH_AMD_DIRECTORY=1 $P r05_prof_c3_directory --workload c3 --queries 100000000 2>&1 | tail -1 | cut -c1-300
SSHASH_AMD_SKTABLE=0 SSHASH_AMD_DIRECTORY=0 $P r05_prof_c3_mphf --workload c3 --queries 100000000 2>&1 | tail -1 | cut -c1-300
$P r05_prof_c3_canonical --workload c3 --canonical 2>&1 | tail -1 | cut -c1-300
