#!/bin/bash
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_eighth; mkdir -p $out
( time timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_switches.py tests/test_gpu_baseline_workloads.py tests/test_gpu_streaming.py tests/test_gpu_km_sweep.py tests/test_gpu_reference_data.py tests/test_sharded.py -x -q -m gpu ) 2>&1 | tail -6 | tee $out/pytest.txt
B="python bench.py --no-cpu-baseline --no-extra-mixes --no-file-query --no-other-paths --no-other-workloads --steps 10 --warmup 2"
val() { python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']/1e9,2), r['ms_per_step'], r['config']['device_bytes_per_kmer'], r['config']['device_stats']['sk_deferred_keys'])"; }
( cp sshash_amd/libsshash_amd.so /tmp/new.so
  for round in 1 2 3; do
    cp tools/debug/libsshash_amd_old.so sshash_amd/libsshash_amd.so; echo -n "c3 old (k-mers' region: 32-byte slots): "; $B --workload c3 2>/dev/null | val
    cp /tmp/new.so sshash_amd/libsshash_amd.so;                      echo -n "c3 new (three 20-byte entries a line): "; $B --workload c3 2>/dev/null | val
  done
  for spk in 2.25 2.0; do for skm in 2.0 1.75; do
    echo -n "c3 new, slots per key $spk, per heavy k-mer $skm: "; SSHASH_AMD_SK_SLOTS_PER_KEY=$spk SSHASH_AMD_SK_SLOTS_PER_KMER=$skm $B --workload c3 2>/dev/null | val
  done; done
  echo -n "c2 old: "; cp tools/debug/libsshash_amd_old.so sshash_amd/libsshash_amd.so; $B --workload c2 2>/dev/null | val
  echo -n "c2 new: "; cp /tmp/new.so sshash_amd/libsshash_amd.so; $B --workload c2 2>/dev/null | val
  echo -n "c2 old: "; cp tools/debug/libsshash_amd_old.so sshash_amd/libsshash_amd.so; $B --workload c2 2>/dev/null | val
  echo -n "c2 new: "; cp /tmp/new.so sshash_amd/libsshash_amd.so; $B --workload c2 2>/dev/null | val
) 2>&1 | tee $out/compact_entries_ab.txt
