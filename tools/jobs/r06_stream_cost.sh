#!/bin/bash
# round 6: what each part of the streaming turn costs at an unchanged number of turns -- builds that execute ONE part twice (on other inputs, result
# kept alive; tools/debug/stream_cost_of_parts.patch, -DSSHASH_STREAM_COST_<part>) against the shipped build, same box: T(twice) - T(shipped)
cd "$(dirname "$0")/../.."
out=gpurun_out/${1:-r06_stream_cost}; mkdir -p $out
export TMPDIR=/tmp SSHASH_BENCH_CACHE=/tmp
for set in "c3 0.95" "c4 0.5" "c3 0.0"; do
  for lib in "" $(ls tools/ab/libvariant_*.so) ""; do
    SSHASH_AMD_LIBRARY=${lib:+$PWD/$lib} python tools/debug/stream_ablation.py $set 2>> $out/err.txt | tee -a $out/cost.txt
  done
done
