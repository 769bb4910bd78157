#!/bin/bash
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_eleventh; mkdir -p $out
( time timeout 1500 python -m pytest tests/test_gpu_streaming.py tests/test_gpu_km_sweep.py tests/test_gpu_reference_data.py -x -q -m gpu ) 2>&1 | tail -5 | tee $out/pytest.txt
cp sshash_amd/libsshash_amd.so /tmp/new.so
run() { python bench.py --streaming --reads 20000000 --steps 5 --warmup 1 --stream-oracle-reads 20000 "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']/1e9,2), r['ms_per_step'], r['config']['positive_fraction_of_kmers'], r['config']['extensions_per_search'])"; }
( for args in "--workload c3 --bases 1000000000 --positive 0.95" "--workload c3 --bases 1000000000 --positive 0.0" "--workload c2 --positive 0.9" "--workload c4" "--workload c4 --positive 0.95"; do
    for round in 1 2; do
      cp tools/debug/libsshash_amd_old.so sshash_amd/libsshash_amd.so; echo -n "$args old (lockstep, inner loop): "; run $args
      cp /tmp/new.so sshash_amd/libsshash_amd.so;                      echo -n "$args new (waiting seeds served every 4th step): "; run $args
    done
    for m in 1 7; do echo -n "$args new, service mask $m: "; SSHASH_AMD_STREAM_SERVICE=$m run $args; done
  done ) 2>&1 | tee $out/ab.txt
