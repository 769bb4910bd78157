#!/bin/bash
# What the driver runs at round end, then the profile of the bench command at this commit.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r03_final
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.jsonl 2> $OUT/bench_driver.err; cut -c1-200 $OUT/bench_driver.jsonl
bash tools/jobs/r03_profile.sh r03_final_prof > $OUT/profile.log 2>&1; tail -12 $OUT/profile.log | cut -c1-300
