#!/bin/bash
# round 6: the streaming kernel with the heavy key remembered -- parity first (streaming tests, the full-size C4 counters), then same-box A/B
# against round 5's turn (tools/ab/libvariant_run_after_seed.so: one event a turn, no memory of heavy keys), then the lane statistics
cd "$(dirname "$0")/../.."
out=gpurun_out/${1:-r06_stream_check3}; mkdir -p $out
export TMPDIR=/tmp SSHASH_BENCH_CACHE=/tmp
timeout 2400 python -m pytest tests/test_gpu_streaming.py tests/test_gpu_reference_data.py "tests/test_gpu_baseline_workloads.py::test_streaming_query_against_the_full_size_k63_dictionary" "tests/test_gpu_baseline_workloads.py::test_streaming_query_at_the_read_count_of_config_c4" -m gpu -x -q > $out/pytest_streaming.txt 2>&1; tail -4 $out/pytest_streaming.txt
for round in 1 2; do
  for set in "c3 0.95" "c4 0.5" "c3 0.0"; do
    for lib in "" $(ls tools/ab/libvariant_*.so); do
      SSHASH_AMD_LIBRARY=${lib:+$PWD/$lib} python tools/debug/stream_ablation.py $set 2>> $out/err.txt | tee -a $out/ab.txt
    done
  done
done
for set in "c3 0.95" "c4 0.5"; do
  for lib in $(ls tools/ab_stats/libstats_*.so); do
    SSHASH_AMD_LIBRARY=$PWD/$lib python tools/debug/stream_stats.py $set 2>> $out/err.txt | tee -a $out/stats.txt
  done
done
