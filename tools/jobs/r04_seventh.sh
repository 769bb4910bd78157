#!/bin/bash
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_seventh; mkdir -p $out
B="python bench.py --no-cpu-baseline --no-extra-mixes --no-file-query --no-other-paths --no-other-workloads --steps 10 --warmup 2"
val() { python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']/1e9,2), r['ms_per_step'], r['config']['device_bytes_per_kmer'])"; }
( for round in 1 2 3; do
    echo -n "c4 resume pass (INWAVE=0): "; SSHASH_AMD_INWAVE=0 $B --workload c4 2>/dev/null | val
    echo -n "c4 finished in the wave:   "; $B --workload c4 2>/dev/null | val
  done ) 2>&1 | tee $out/inwave_k63_ab.txt
( cp sshash_amd/libsshash_amd.so /tmp/new.so
  for round in 1 2 3; do
    cp tools/debug/libsshash_amd_old.so sshash_amd/libsshash_amd.so; echo -n "c3 old (before the streaming change; headline kernel at 72 registers by the pad guard): "; $B --workload c3 2>/dev/null | val
    cp /tmp/new.so sshash_amd/libsshash_amd.so;                      echo -n "c3 new (registers renamed by the guard): "; $B --workload c3 2>/dev/null | val
  done ) 2>&1 | tee $out/c3_ab.txt
NAME=r04_seventh bash tools/jobs/r04_ab_stream.sh
( time timeout 2400 python -m pytest tests/test_gpu_switches.py tests/test_gpu_parity.py tests/test_gpu_baseline_workloads.py tests/test_gpu_km_sweep.py tests/test_gpu_reference_data.py -x -q -m gpu ) 2>&1 | tail -6 | tee $out/pytest.txt
