#!/bin/bash
# the whole GPU suite + smoke, as the driver runs them
cd "$(dirname "$0")/../.."
out=gpurun_out/${1:-r05_suite}; mkdir -p $out
( time timeout 3300 python -m pytest tests -x -q -m gpu ) 2>&1 | tail -40 | tee $out/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $out/smoke.txt
