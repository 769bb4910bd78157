#!/bin/bash
# same-box A/B of the streaming kernel: tools/debug/libsshash_amd_old.so against sshash_amd/libsshash_amd.so; high-hit (90 % of the reads from the dictionary) and low-hit
cd "$(dirname "$0")/../.."
out=gpurun_out/${NAME:-r04_ab_stream}; mkdir -p $out
cp sshash_amd/libsshash_amd.so /tmp/new.so
run() { python bench.py --streaming --reads ${READS:-20000000} --steps 5 --warmup 1 --stream-oracle-reads 20000 "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']/1e9,2), r['ms_per_step'], r['config']['positive_fraction_of_kmers'], r['config']['extensions_per_search'])"; }
for wl in c2 c4; do
  for pos in 0.9 0.0; do
    for round in 1 2; do
      cp tools/debug/libsshash_amd_old.so sshash_amd/libsshash_amd.so; echo -n "$wl positive $pos old: "; run --workload $wl --positive $pos
      cp /tmp/new.so sshash_amd/libsshash_amd.so;                      echo -n "$wl positive $pos new: "; run --workload $wl --positive $pos
    done
  done
done 2>&1 | tee $out/ab.txt
timeout 900 python -m pytest tests/test_gpu_streaming.py -x -q -m gpu 2>&1 | tail -4 | tee $out/pytest.txt
