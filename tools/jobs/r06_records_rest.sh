#!/bin/bash
# round 6: the PMC records of the lines whose kernels did NOT change this round, re-made at this round's commit all the same
cd "$(dirname "$0")/../.."
export SSHASH_BENCH_CACHE=/tmp
P="bash tools/jobs/r06_profile.sh"
for w in c2 c4; do $P r06_prof_$w --workload $w 2>&1 | tail -1 | cut -c1-300; done
SSHASH_AMD_SKTABLE=0 SSHASH_AMD_DIRECTORY=1 $P r06_prof_c3_directory --workload c3 --queries 100000000 2>&1 | tail -1 | cut -c1-300
SSHASH_AMD_SKTABLE=0 SSHASH_AMD_DIRECTORY=0 $P r06_prof_c3_mphf --workload c3 --queries 100000000 2>&1 | tail -1 | cut -c1-300
$P r06_prof_c3_canonical --workload c3 --canonical 2>&1 | tail -1 | cut -c1-300
