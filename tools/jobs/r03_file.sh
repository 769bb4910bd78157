#!/bin/bash
# directory path with both strands resolved at once (parity + other_paths), then the end-to-end FASTQ query
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r03_file
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_sharded.py -x -q -m gpu > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
python bench.py --no-cpu-baseline --no-extra-mixes --steps 5 --warmup 2 > $OUT/bench.jsonl 2> $OUT/bench.err
python3 -c "
import json
r=json.loads(open('$OUT/bench.jsonl').read().strip().splitlines()[-1]); print(r['value'], json.dumps(r['other_paths']))"
df -h /tmp | tail -1; free -g | head -2
timeout 1500 python tools/bench_streaming_file.py --reads ${1:-20000000} > $OUT/file_${1:-20000000}.jsonl 2> $OUT/file.err
tail -5 $OUT/file.err; cat $OUT/file_${1:-20000000}.jsonl | cut -c1-2500
