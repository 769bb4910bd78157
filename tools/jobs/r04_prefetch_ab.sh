#!/bin/bash
# same-box A/B: the key's bucket fetched before the run is tried (SSHASH_AMD_STREAM_PREFETCH=1) against the default
cd "$(dirname "$0")/../.."
out=gpurun_out/${NAME:-r04_prefetch_ab}; mkdir -p $out
SSHASH_AMD_STREAM_PREFETCH=1 timeout 900 python -m pytest tests/test_gpu_streaming.py -x -q -m gpu 2>&1 | tail -3 | tee $out/pytest.txt
run() { python bench.py --streaming --reads 20000000 --steps 5 --warmup 1 --stream-oracle-reads 20000 "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']/1e9,2), r['ms_per_step'], r['config'].get('positive_fraction_of_kmers'), r['config'].get('extensions_per_search'))"; }
{
for pos in 0.95 0.0; do
  for round in 1 2; do
    echo -n "k31 human stand-in 1e9 bases, positive $pos, default:  "; run --workload c3 --bases 1000000000 --positive $pos
    echo -n "k31 human stand-in 1e9 bases, positive $pos, prefetch: "; SSHASH_AMD_STREAM_PREFETCH=1 run --workload c3 --bases 1000000000 --positive $pos
  done
done
for pos in 0.5 0.95; do
  echo -n "c4 (k=63), positive $pos, default:  "; run --workload c4 --positive $pos
  echo -n "c4 (k=63), positive $pos, prefetch: "; SSHASH_AMD_STREAM_PREFETCH=1 run --workload c4 --positive $pos
done
} 2>&1 | tee $out/ab.txt
