#!/bin/bash
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_alltests; mkdir -p $out
( time timeout 3300 python -m pytest tests -x -q -m gpu ) 2>&1 | tail -12 | tee $out/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $out/smoke.txt
