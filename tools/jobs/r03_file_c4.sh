#!/bin/bash
# Round 3: BASELINE config C4 end to end -- the human k = 63 stand-in, 10^8 reads x 150 bp (half drawn from it, half random) as a FASTQ file
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03_file_c4
timeout 600 python bench.py --workload c4 --bases 300000000 --queries 20000000 --no-cpu-baseline --no-other-paths --no-extra-mixes --steps 3 --warmup 1 2>gpurun_out/r03_file_c4/small.err | python3 -c "
import json,sys; r=json.loads(sys.stdin.read()); print('small c4 bench:', r['value']/1e9, json.dumps(r['streaming_from_file'])[:900])"
tail -3 gpurun_out/r03_file_c4/small.err | cut -c1-300
SSHASH_AMD_VERBOSE=1 timeout 3000 python tools/bench_streaming_file.py --c4 --reads 100000000 > gpurun_out/r03_file_c4/file_1e8.jsonl 2> gpurun_out/r03_file_c4/file_1e8.err
tail -12 gpurun_out/r03_file_c4/file_1e8.err | cut -c1-300; cut -c1-1500 gpurun_out/r03_file_c4/file_1e8.jsonl
