#!/bin/bash
# Round 3, last kernel change: the whole GPU suite, then what the driver runs + the profile of the bench command, the C2 / C4 lines, sharded N = 1
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03_final
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r03_final/pytest.log 2>&1; tail -3 gpurun_out/r03_final/pytest.log
bash tools/jobs/r03_final.sh
OUT=gpurun_out/r03_final
timeout 900 python bench.py --workload c2 > $OUT/bench_c2.jsonl 2> $OUT/bench_c2.err; cut -c1-160 $OUT/bench_c2.jsonl
timeout 900 python bench.py --workload c2 --canonical --no-cpu-baseline --no-file-query > $OUT/bench_c2_canonical.jsonl 2> $OUT/bench_c2_canonical.err; cut -c1-160 $OUT/bench_c2_canonical.jsonl
timeout 1200 python bench.py --workload c4 --no-cpu-baseline --no-file-query > $OUT/bench_c4.jsonl 2> $OUT/bench_c4.err; cut -c1-160 $OUT/bench_c4.jsonl
timeout 900 python bench.py --workload c2 --sharded table --no-cpu-baseline --no-extra-mixes --no-other-paths --no-file-query > $OUT/bench_sharded_table.jsonl 2> $OUT/bench_sharded_table.err; cut -c1-160 $OUT/bench_sharded_table.jsonl
timeout 900 python bench.py --workload c2 --sharded minimizer --no-cpu-baseline --no-extra-mixes --no-other-paths --no-file-query > $OUT/bench_sharded_minimizer.jsonl 2> $OUT/bench_sharded_minimizer.err; cut -c1-160 $OUT/bench_sharded_minimizer.jsonl
