#!/bin/bash
# after the table's key length became its own parameter (k > 31: at least k - 32): the GPU suite, smoke, the driver's command,
# the C4 profile (kernel stats + PMC -> profiles/traffic.json["c4"]), the sliding election at k = 63 once more (its columns are shorter now)
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_final4; mkdir -p $out
( time timeout 3300 python -m pytest tests -x -q -m gpu ) 2>&1 | tail -8 | tee $out/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $out/smoke.txt
bash tools/jobs/r04_driver_command.sh 2>&1 | tail -14 | tee $out/driver_command.txt
sval() { python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']/1e9,2), r['ms_per_step'], r['config'].get('positive_fraction_of_kmers'))"; }
( for round in 1 2; do
    for pos in 0.5 0.0; do
      for roll in 0 1; do echo -n "c4 streaming, positive $pos, sliding election $roll: "; SSHASH_AMD_STREAM_ROLLING=$roll python bench.py --streaming --reads 20000000 --steps 5 --warmup 1 --stream-oracle-reads 20000 --workload c4 --positive $pos 2>/dev/null | sval; done
    done
  done ) 2>&1 | tee $out/streaming_rolling_k63_table_m31_ab.txt
bash tools/jobs/r04_profile.sh r04_prof_c4 --workload c4 2>&1 | tail -4 | cut -c1-600
