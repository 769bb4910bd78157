#!/bin/bash
# same-box A/B of the rolling key election at k = 63 (n = 39: 80 KB of LDS a workgroup, two workgroups a CU)
cd "$(dirname "$0")/../.."
out=gpurun_out/${NAME:-r04_roll_ab2}; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_streaming.py -x -q -m gpu 2>&1 | tail -4 | tee $out/pytest.txt
run() { python bench.py --streaming --reads 20000000 --steps 5 --warmup 1 --stream-oracle-reads 20000 "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']/1e9,2), r['ms_per_step'], r['config'].get('positive_fraction_of_kmers'), r['config'].get('extensions_per_search'))"; }
{
for pos in 0.5 0.95 0.0; do
  for round in 1 2; do
    echo -n "c4 (k=63 m=25), positive $pos, from scratch: "; SSHASH_AMD_STREAM_ROLLING=0 run --workload c4 --positive $pos
    echo -n "c4 (k=63 m=25), positive $pos, rolling:      "; run --workload c4 --positive $pos
  done
done
} 2>&1 | tee $out/ab.txt
