#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03_file2
python bench.py --no-cpu-baseline --no-extra-mixes --no-other-paths --no-file-query --steps 2 --warmup 1 > /dev/null 2>&1
timeout 1500 python tools/bench_streaming_file.py --reads ${1:-20000000} > gpurun_out/r03_file2/file_${1:-20000000}.jsonl 2> gpurun_out/r03_file2/file.err
python3 -c "
import json
r=json.loads(open('gpurun_out/r03_file2/file_${1:-20000000}.jsonl').read())
for f in ('fastq','fastq.gz'): print(f, r[f]['seconds'], r[f]['ns_per_kmer'], r[f]['reader_alone']['seconds'])"
