#!/bin/bash
# round 6: the staging area's regions padded to 65 pieces (no LDS bank conflict in the slot examinations) against 64, same box, alternating:
# parity first, then the streaming sets, then the lookup lines (several processes each: a replica draws its memory's rate per allocation)
cd "$(dirname "$0")/../.."
out=gpurun_out/${1:-r06_stage_padding_ab}; mkdir -p $out
export TMPDIR=/tmp SSHASH_BENCH_CACHE=/tmp
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_streaming.py tests/test_gpu_reference_data.py tests/test_gpu_km_sweep.py -m gpu -x -q > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
for round in 1 2; do for set in "c3 0.95" "c4 0.5" "c3 0.0"; do for lib in "" $(ls tools/ab/libvariant_*.so); do
  SSHASH_AMD_LIBRARY=${lib:+$PWD/$lib} python tools/debug/stream_ablation.py $set 2>> $out/err.txt | tee -a $out/streaming.txt
done; done; done
S="--steps 10 --warmup 3 --no-cpu-baseline --no-extra-mixes --no-file-query --no-other-paths --no-other-workloads --no-line-probe --quiet-record"
for round in 1 2 3; do for w in c4 c3 c2; do for lib in "" $(ls tools/ab/libvariant_*.so); do
  tag=${w}_${round}_$(basename ${lib:-padded_65} .so)
  SSHASH_AMD_LIBRARY=${lib:+$PWD/$lib} python bench.py $S --workload $w --full-record $out/$tag.json > $out/$tag.jsonl 2>> $out/bench.err
  python3 -c "
import json; r=json.load(open('$out/$tag.json')); print('$tag', round(r['value']/1e9,2), 'G lookups/s', r['ms_per_step'], 'ms/step')" | tee -a $out/lookups.txt
done; done; done
