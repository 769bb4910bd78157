#!/bin/bash
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_fifth; mkdir -p $out
timeout 900 tools/debug/vgpr64_check 2>&1 | tee $out/vgpr64_check.jsonl
timeout 1200 python -m pytest tests/test_gpu_streaming.py -x -q -m gpu -k "file or fastq" 2>&1 | tail -15 | tee $out/pytest_streaming_files.txt
timeout 900 python tools/bench_streaming_file.py --reads 20000000 --bases 1387536274 2>&1 | tail -3 | tee $out/file_query_2e7.jsonl
