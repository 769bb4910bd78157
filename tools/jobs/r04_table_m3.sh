#!/bin/bash
# k = 63 (C4): longer table keys -- fewer k-mers under heavy keys, fewer candidates to elect; same-box alternating; the streaming query too
cd "$(dirname "$0")/../.."
out=gpurun_out/${NAME:-r04_table_m3}; mkdir -p $out
B="python bench.py --no-cpu-baseline --no-extra-mixes --no-file-query --no-other-workloads --no-other-paths --steps 10 --warmup 2"
val() { python -c "import json,sys; r=json.loads(sys.stdin.read()); c=r['config']; print(round(r['value']/1e9,2), r['ms_per_step'], c['device_bytes_per_kmer'], (c.get('table_histogram') or {}).get('super_kmers'), (c.get('table_histogram') or {}).get('kmers_under_heavy_keys'))"; }
sval() { python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']/1e9,2), r['ms_per_step'])"; }
( for round in 1 2 3; do
    for m in 25 29 31 32; do echo -n "c4 table m $m: "; SSHASH_AMD_SK_M=$m $B --workload c4 2>/dev/null | val; done
  done
  for m in 31 32; do echo -n "c4 table m $m, 3.0 slots per key: "; SSHASH_AMD_SK_SLOTS_PER_KEY=3.0 SSHASH_AMD_SK_M=$m $B --workload c4 2>/dev/null | val; done
  for round in 1 2; do
    for m in 25 29 31; do echo -n "c4 streaming, table m $m: "; SSHASH_AMD_SK_M=$m python bench.py --streaming --reads 20000000 --steps 5 --warmup 1 --stream-oracle-reads 20000 --workload c4 2>/dev/null | sval; done
  done
) 2>&1 | tee $out/table_m_sweep_c4.txt
for m in 31 32; do
  echo "== SSHASH_AMD_SK_M=$m (k = 63 tests)"; SSHASH_AMD_SK_M=$m timeout 1500 python -m pytest tests/test_gpu_km_sweep.py tests/test_gpu_reference_data.py tests/test_gpu_streaming.py -q -m gpu 2>&1 | tail -3
done 2>&1 | tee $out/pytest.txt
