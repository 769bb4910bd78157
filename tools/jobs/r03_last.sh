#!/bin/bash
# Round 3, last commit: the GPU suite twice, then what the driver runs
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r03_last; mkdir -p $OUT
for i in 1 2; do timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/pytest_$i.log 2>&1; grep -v "^Extension" $OUT/pytest_$i.log | tail -2; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.jsonl 2> $OUT/bench_driver.err; cut -c1-220 $OUT/bench_driver.jsonl
