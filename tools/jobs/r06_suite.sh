#!/bin/bash
# round 6: what the driver runs at round end -- pytest -m gpu, smoke -- on a fresh box
cd "$(dirname "$0")/../.."
out=gpurun_out/${1:-r06_suite}; mkdir -p $out
export TMPDIR=/tmp
timeout 3000 python -m pytest tests/ -x -q -m gpu --durations=15 > $out/pytest_gpu.txt 2>&1; tail -22 $out/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; tail -2 $out/smoke.txt
