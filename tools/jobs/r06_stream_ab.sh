#!/bin/bash
# round 6: same-box A/B of builds of the streaming kernel (how every streaming_*_ab.txt of profiles/r06 was made; the lane statistics:
# r06_stream_stats.sh over tools/ab_stats/) -- the shipped library against tools/ab/libvariant_*.so (built with
# `make -C sshash_amd/csrc DEFS=-D...`, copied aside, default rebuilt) -- on the bench's read sets, alternating, counters printed (they must agree).
cd "$(dirname "$0")/../.."
out=gpurun_out/${1:-r06_stream_ab}; mkdir -p $out
export TMPDIR=/tmp SSHASH_BENCH_CACHE=/tmp
timeout 1500 python -m pytest tests/test_gpu_streaming.py -m gpu -x -q > $out/pytest_streaming.txt 2>&1; tail -3 $out/pytest_streaming.txt
for round in 1 2; do
  for set in "c3 0.95" "c4 0.5" "c3 0.0"; do
    for lib in "" $(ls tools/ab/libvariant_*.so); do
      SSHASH_AMD_LIBRARY=${lib:+$PWD/$lib} python tools/debug/stream_ablation.py $set 2>> $out/err.txt | tee -a $out/ab.txt
    done
  done
done
