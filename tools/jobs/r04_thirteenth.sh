#!/bin/bash
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_thirteenth; mkdir -p $out
cp sshash_amd/libsshash_amd.so /tmp/default.so
B="python bench.py --no-cpu-baseline --no-extra-mixes --no-file-query --no-other-paths --no-other-workloads --steps 10 --warmup 2"
val() { python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']/1e9,2), r['ms_per_step'])"; }
( for round in 1 2 3 4; do
    cp /tmp/default.so sshash_amd/libsshash_amd.so; echo -n "c4 heavy hop only: "; $B --workload c4 2>/dev/null | val
    cp tools/debug/libsshash_amd_second.so sshash_amd/libsshash_amd.so; echo -n "c4 + one look at the key's second choice: "; $B --workload c4 2>/dev/null | val
  done ) 2>&1 | tee $out/second_choice_ab.txt
( for round in 1 2; do
    cp tools/debug/libsshash_amd_old.so sshash_amd/libsshash_amd.so; echo -n "ascii old: "; python tools/perf_variants.py --skip-streaming 2>/dev/null | grep -E '"ascii_ids_mix50"|"packed_ids_mix50"' | cut -c1-200 | tr '\n' ' '; echo
    cp /tmp/default.so sshash_amd/libsshash_amd.so; echo -n "ascii new: "; python tools/perf_variants.py --skip-streaming 2>/dev/null | grep -E '"ascii_ids_mix50"|"packed_ids_mix50"' | cut -c1-200 | tr '\n' ' '; echo
  done ) 2>&1 | tee $out/ascii_ab.txt
cp /tmp/default.so sshash_amd/libsshash_amd.so
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_km_sweep.py -x -q -m gpu ) 2>&1 | tail -4 | tee $out/pytest.txt
