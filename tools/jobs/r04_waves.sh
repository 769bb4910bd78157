#!/bin/bash
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_waves; mkdir -p $out
B="python bench.py --no-cpu-baseline --no-extra-mixes --no-file-query --no-other-paths --no-other-workloads --steps 10 --warmup 2"
val() { python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']/1e9,2), r['ms_per_step'])"; }
( for round in 1 2 3 4; do
    cp tools/debug/libsshash_amd_old.so sshash_amd/libsshash_amd.so;    echo -n "c4 66 registers (7 waves): "; $B --workload c4 2>/dev/null | val
    cp tools/debug/libsshash_amd_w8wide.so sshash_amd/libsshash_amd.so; echo -n "c4 64 registers (8 waves): "; $B --workload c4 2>/dev/null | val
  done
  for round in 1 2 3 4; do
    cp tools/debug/libsshash_amd_old.so sshash_amd/libsshash_amd.so;   echo -n "c3 67 registers (7 waves): "; $B --workload c3 2>/dev/null | val
    cp tools/debug/libsshash_amd_w8all.so sshash_amd/libsshash_amd.so; echo -n "c3 64 registers, 12 bytes of scratch (8 waves): "; $B --workload c3 2>/dev/null | val
  done ) 2>&1 | tee $out/ab.txt
cp tools/debug/libsshash_amd_w8all.so sshash_amd/libsshash_amd.so
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_km_sweep.py tests/test_gpu_switches.py tests/test_gpu_reference_data.py -x -q -m gpu ) 2>&1 | tail -6 | tee $out/pytest_w8all.txt
