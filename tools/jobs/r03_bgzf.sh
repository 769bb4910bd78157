#!/bin/bash
# Round 3: the file query on BGZF input (members inflated on all cores) next to plain and gzip, 10^8 reads on the C3 stand-in
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r03_bgzf; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_bench_harness.py tests/test_gpu_streaming.py -x -q -m gpu 2>&1 | tail -2
python bench.py --no-cpu-baseline --no-extra-mixes --no-other-paths --no-file-query --steps 2 --warmup 1 > /dev/null 2>&1
SSHASH_AMD_VERBOSE=1 timeout 3000 python tools/bench_streaming_file.py --reads 100000000 > $OUT/file_1e8.jsonl 2> $OUT/file_1e8.err
grep "file query\|written" $OUT/file_1e8.err | cut -c1-200
python3 -c "
import json
r=json.loads(open('$OUT/file_1e8.jsonl').read().strip().splitlines()[-1])
for f in ('fastq','fastq.gz','bgzf.fastq.gz'): print(f, r[f]['seconds'], r[f]['ns_per_kmer'], r[f].get('reader_alone'))
"
