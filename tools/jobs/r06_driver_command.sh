#!/bin/bash
# the driver's command, as the driver runs it: ONE compact stdout line (< 8 KB), the full record in bench_full.json
cd "$(dirname "$0")/../.."
out=gpurun_out/${1:-r06_driver_command}; mkdir -p $out
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 --full-record $out/bench_full.json > $out/bench.jsonl 2> $out/bench.err ) 2>&1 | tail -3
grep -v "full record: {" $out/bench.err | tail -c 700; wc -c $out/bench.jsonl; cat $out/bench.jsonl
