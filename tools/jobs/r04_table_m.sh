#!/bin/bash
# the table's own key length (sk_view::m, SSHASH_AMD_SK_M): parity under other lengths, then the same-box sweep on C3 / C2 / C4
cd "$(dirname "$0")/../.."
out=gpurun_out/${NAME:-r04_table_m}; mkdir -p $out
for m in 17 14; do
  echo "== SSHASH_AMD_SK_M=$m"; SSHASH_AMD_SK_M=$m timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_streaming.py tests/test_gpu_km_sweep.py tests/test_sharded.py tests/test_gpu_switches.py -x -q -m gpu 2>&1 | tail -4
done 2>&1 | tee $out/pytest.txt
B="python bench.py --no-cpu-baseline --no-extra-mixes --no-file-query --no-other-workloads --no-other-paths --steps 10 --warmup 2"
val() { python -c "import json,sys; r=json.loads(sys.stdin.read()); c=r['config']; print(round(r['value']/1e9,2), r['ms_per_step'], c['device_bytes_per_kmer'], (c.get('table_histogram') or {}).get('super_kmers'), (c.get('table_histogram') or {}).get('kmers_under_heavy_keys'))"; }
sval() { python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']/1e9,2), r['ms_per_step'])"; }
( for round in 1 2; do
    for m in 21 19 17 15; do echo -n "c3 table m $m: "; SSHASH_AMD_SK_M=$m $B --workload c3 2>/dev/null | val; done
  done
  for m in 21 19 17; do echo -n "c2 table m $m: "; SSHASH_AMD_SK_M=$m $B --workload c2 2>/dev/null | val; done
  for m in 25 21 19; do echo -n "c4 table m $m: "; SSHASH_AMD_SK_M=$m $B --workload c4 2>/dev/null | val; done
  for m in 21 19 17; do echo -n "c3 streaming high-hit, table m $m: "; SSHASH_AMD_SK_M=$m python bench.py --streaming --reads 20000000 --steps 5 --warmup 1 --stream-oracle-reads 20000 --workload c3 --positive 0.95 2>/dev/null | sval; done
) 2>&1 | tee $out/table_m_sweep.txt
