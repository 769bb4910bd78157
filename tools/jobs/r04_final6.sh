#!/bin/bash
# after the k-mers' region at k > 31 went to 2.5 places per k-mer (31-base key, 3.0 slots per item): the GPU suite, smoke, the driver's command,
# the C4 profile (kernel stats + PMC -> profiles/traffic.json["c4"])
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_final6; mkdir -p $out
( time timeout 3300 python -m pytest tests -x -q -m gpu ) 2>&1 | tail -8 | tee $out/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $out/smoke.txt
bash tools/jobs/r04_driver_command.sh 2>&1 | tail -14 | tee $out/driver_command.txt
bash tools/jobs/r04_profile.sh r04_prof_c4 --workload c4 2>&1 | tail -1 | cut -c1-300
