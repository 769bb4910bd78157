#!/bin/bash
# same-box A/B: the read's characters loaded eight bases ahead (k <= 31); tools/debug/libsshash_amd_old.so = the commit before
cd "$(dirname "$0")/../.."
out=gpurun_out/${NAME:-r04_ahead_ab}; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_streaming.py -x -q -m gpu 2>&1 | tail -3 | tee $out/pytest.txt
run() { python bench.py --streaming --reads 20000000 --steps 5 --warmup 1 --stream-oracle-reads 20000 "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']/1e9,2), r['ms_per_step'], r['config'].get('positive_fraction_of_kmers'), r['config'].get('extensions_per_search'))"; }
{
for pos in 0.95 0.0; do
  for round in 1 2; do
    echo -n "k31 human stand-in 1e9 bases, positive $pos, before: "; SSHASH_AMD_LIBRARY=$PWD/tools/debug/libsshash_amd_old.so run --workload c3 --bases 1000000000 --positive $pos
    echo -n "k31 human stand-in 1e9 bases, positive $pos, ahead:  "; run --workload c3 --bases 1000000000 --positive $pos
  done
done
echo -n "c2 positive 0.9, before: "; SSHASH_AMD_LIBRARY=$PWD/tools/debug/libsshash_amd_old.so run --workload c2 --positive 0.9
echo -n "c2 positive 0.9, ahead:  "; run --workload c2 --positive 0.9
echo -n "c4 (k=63), positive 0.5, before: "; SSHASH_AMD_LIBRARY=$PWD/tools/debug/libsshash_amd_old.so run --workload c4 --positive 0.5
echo -n "c4 (k=63), positive 0.5, now:    "; run --workload c4 --positive 0.5
} 2>&1 | tee $out/ab.txt
