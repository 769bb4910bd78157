#!/bin/bash
# `sshash query` end to end on a FASTQ of 2 x 10^7 reads (plain / gzip / BGZF), k = 31 high-hit and config C4's set
cd "$(dirname "$0")/../.."
out=gpurun_out/r05_file_query; mkdir -p $out
bash tools/jobs/r05_stream_check.sh r05_stream_v7 2>&1 | tail -8
python tools/bench_streaming_file.py --reads 20000000 2> $out/k31.err | tail -1 > $out/k31.jsonl; cut -c1-1500 $out/k31.jsonl
python tools/bench_streaming_file.py --reads 20000000 --c4 2> $out/c4.err | tail -1 > $out/c4.jsonl; cut -c1-1500 $out/c4.jsonl
