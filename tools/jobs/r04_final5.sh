#!/bin/bash
# after k > 31 went to 3.0 slots per item (with the 31-base table key): the GPU suite, smoke, the driver's command, the C4 profile
# (kernel stats + PMC -> profiles/traffic.json["c4"]), C4's streaming query A/B against 2.5 slots
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_final5; mkdir -p $out
( time timeout 3300 python -m pytest tests -x -q -m gpu ) 2>&1 | tail -8 | tee $out/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $out/smoke.txt
bash tools/jobs/r04_driver_command.sh 2>&1 | tail -14 | tee $out/driver_command.txt
sval() { python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']/1e9,2), r['ms_per_step'], r['config'].get('positive_fraction_of_kmers'))"; }
( for round in 1 2; do
    for s in 2.5 3.0; do echo -n "c4 streaming, $s slots per key: "; SSHASH_AMD_SK_SLOTS_PER_KEY=$s python bench.py --streaming --reads 20000000 --steps 5 --warmup 1 --stream-oracle-reads 20000 --workload c4 2>/dev/null | sval; done
  done ) 2>&1 | tee $out/streaming_c4_slots_ab.txt
bash tools/jobs/r04_profile.sh r04_prof_c4 --workload c4 2>&1 | tail -1 | cut -c1-300
