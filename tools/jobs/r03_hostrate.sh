#!/bin/bash
cd "$(dirname "$0")/../.."
python bench.py --no-cpu-baseline --no-extra-mixes --no-other-paths --no-file-query --steps 2 --warmup 1 > /dev/null 2>&1
python tools/debug/stream_host_rate.py 20000000 2>&1 | tail -4
