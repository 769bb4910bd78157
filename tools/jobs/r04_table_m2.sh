#!/bin/bash
# the table's own key length, second job: the rest of the parity files under other lengths (the first job stopped at a statistics
# assertion of test_accelerators_disabled[sktable_packed_tight], which expects the default key's heavy k-mers), and longer keys on C3
cd "$(dirname "$0")/../.."
out=gpurun_out/${NAME:-r04_table_m2}; mkdir -p $out
for m in 17 14 25; do
  echo "== SSHASH_AMD_SK_M=$m"; SSHASH_AMD_SK_M=$m timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_streaming.py tests/test_gpu_km_sweep.py tests/test_sharded.py tests/test_gpu_reference_data.py -q -m gpu --deselect "tests/test_gpu_parity.py::test_accelerators_disabled[sktable_packed_tight]" 2>&1 | tail -4
done 2>&1 | tee $out/pytest.txt
timeout 1500 python -m pytest tests/test_gpu_switches.py -q -m gpu -k "table_key or default" 2>&1 | tail -3 | tee -a $out/pytest.txt
B="python bench.py --no-cpu-baseline --no-extra-mixes --no-file-query --no-other-workloads --no-other-paths --steps 10 --warmup 2"
val() { python -c "import json,sys; r=json.loads(sys.stdin.read()); c=r['config']; print(round(r['value']/1e9,2), r['ms_per_step'], c['device_bytes_per_kmer'], (c.get('table_histogram') or {}).get('super_kmers'), (c.get('table_histogram') or {}).get('kmers_under_heavy_keys'))"; }
( for round in 1 2; do
    for m in 21 23 25; do echo -n "c3 table m $m: "; SSHASH_AMD_SK_M=$m $B --workload c3 2>/dev/null | val; done
  done
  for m in 25 29; do echo -n "c4 table m $m: "; SSHASH_AMD_SK_M=$m $B --workload c4 2>/dev/null | val; done
) 2>&1 | tee $out/table_m_sweep.txt
