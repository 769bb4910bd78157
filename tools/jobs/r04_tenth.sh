#!/bin/bash
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_tenth; mkdir -p $out
( time timeout 1500 python -m pytest tests/test_gpu_switches.py tests/test_gpu_parity.py -x -q -m gpu ) 2>&1 | tail -5 | tee $out/pytest.txt
B="python bench.py --no-cpu-baseline --no-extra-mixes --no-file-query --no-other-workloads --steps 10 --warmup 2"
paths() { python -c "import json,sys; r=json.loads(sys.stdin.read()); o=r['other_paths']; print(round(r['value']/1e9,2), {k: (round(v['lookups_per_s']/1e9,2), v['ms']) for k,v in o.items()})"; }
( cp sshash_amd/libsshash_amd.so /tmp/new.so
  for round in 1 2; do
    cp tools/debug/libsshash_amd_old.so sshash_amd/libsshash_amd.so; echo -n "c3 table-less paths, old (a lane reads its own 32-byte units): "; $B --workload c3 2>/dev/null | paths
    cp /tmp/new.so sshash_amd/libsshash_amd.so;                      echo -n "c3 table-less paths, new (pairs of lanes): "; $B --workload c3 2>/dev/null | paths
  done ) 2>&1 | tee $out/tableless_pairs_ab.txt
B="$B --no-other-paths"
val() { python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']/1e9,2), r['ms_per_step'], r['config']['device_bytes_per_kmer'])"; }
( for round in 1 2 3 4; do
    echo -n "c3 2.5 slots per key:  "; $B --workload c3 2>/dev/null | val
    echo -n "c3 2.25 slots per key: "; SSHASH_AMD_SK_SLOTS_PER_KEY=2.25 $B --workload c3 2>/dev/null | val
  done ) 2>&1 | tee $out/slots_per_key_ab.txt
bash tools/jobs/r04_ninth.sh 2>&1 | tee $out/streaming_profiles.txt
