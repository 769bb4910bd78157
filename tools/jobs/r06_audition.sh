#!/bin/bash
# round 6: the table's block auditioned at upload (sktable.hip) against not (table_auditions=1), processes alternating on one box
cd "$(dirname "$0")/../.."
out=gpurun_out/${1:-r06_audition}; mkdir -p $out
export SSHASH_BENCH_CACHE=/tmp SSHASH_AMD_VERBOSE=1
for p in 1 2 3; do
  python tools/debug/upload_modes.py 3 2>> $out/auditioned.err | sed 's/^/auditioned: /' | tee -a $out/modes.txt
  SSHASH_AMD_TEST_HOOKS=table_auditions=1 python tools/debug/upload_modes.py 3 2>> $out/plain.err | sed 's/^/first block: /' | tee -a $out/modes.txt
done
grep "table block" $out/auditioned.err | tail -40 > $out/blocks.txt
