#!/bin/bash
# round 6: lane occupancy of the streaming kernel's turns (debug builds with -DSSHASH_STREAM_STATS, tools/ab_stats/)
cd "$(dirname "$0")/../.."
out=gpurun_out/${1:-r06_stream_stats}; mkdir -p $out
export TMPDIR=/tmp SSHASH_BENCH_CACHE=/tmp
for set in "c3 0.95" "c4 0.5" "c3 0.0"; do
  for lib in $(ls tools/ab_stats/libstats_*.so); do
    SSHASH_AMD_LIBRARY=$PWD/$lib python tools/debug/stream_stats.py $set 2>> $out/err.txt | tee -a $out/stats.txt
  done
done
