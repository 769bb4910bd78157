#!/bin/bash
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r03_bgzf2; mkdir -p $OUT
python bench.py --no-cpu-baseline --no-extra-mixes --no-other-paths --no-file-query --steps 2 --warmup 1 > /dev/null 2>&1
SSHASH_AMD_VERBOSE=1 timeout 3000 python tools/bench_streaming_file.py --reads 100000000 > $OUT/file_1e8.jsonl 2> $OUT/file_1e8.err
grep "file query\|written" $OUT/file_1e8.err | cut -c1-200
python3 -c "
import json
r=json.loads(open('$OUT/file_1e8.jsonl').read().strip().splitlines()[-1])
for f in ('fastq','fastq.gz','bgzf.fastq.gz'): print(f, r[f]['seconds'], r[f]['ns_per_kmer'], r[f].get('reader_alone'))
"
