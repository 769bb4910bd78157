#!/bin/bash
# Round 2: election hash over the first 16 bases of an m-mer (sk_key): parity, then C3, k = 63 and canonical bench lines.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r02_select
mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_km_sweep.py tests/test_gpu_streaming.py tests/test_gpu_baseline_workloads.py -m gpu -x -q > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log
show() { python -c "
import json,sys
r=json.load(open(sys.argv[1])); print(sys.argv[1], round(r['value']/1e9,2), r['ms_per_step'], r['config']['device_stats'].get('sk_slots_used'), r['config']['device_bytes_per_kmer'], {k:round(v['lookups_per_s']/1e9,2) for k,v in (r.get('other_mixes') or {}).items()})" $1; }
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_c3.jsonl 2> $OUT/bench_c3.err; show $OUT/bench_c3.jsonl
timeout 900 python bench.py --workload c2 --k 63 --m 25 --bases 1500000000 --mean-len 160 --no-cpu-baseline --no-extra-mixes > $OUT/bench_k63.jsonl 2>$OUT/bench_k63.err; show $OUT/bench_k63.jsonl
timeout 900 python bench.py --workload c2 --canonical --no-cpu-baseline --no-extra-mixes > $OUT/bench_canonical.jsonl 2>$OUT/bench_canonical.err; show $OUT/bench_canonical.jsonl
