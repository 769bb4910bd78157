#!/bin/bash
# end of round 4, at the last streaming commit (characters loaded ahead): the GPU suite, smoke, the driver's command
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_final3; mkdir -p $out
( time timeout 3300 python -m pytest tests -x -q -m gpu ) 2>&1 | tail -8 | tee $out/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $out/smoke.txt
bash tools/jobs/r04_driver_command.sh 2>&1 | tail -14
