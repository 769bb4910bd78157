#!/bin/bash
# Round 3: compact k-mer entries at k <= 63 -- parity (k = 63 cases, tight region), then C4 before/after is against profiles/r03/bench_c4_final.jsonl
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03_compact
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r03_compact/pytest.log 2>&1; tail -3 gpurun_out/r03_compact/pytest.log
SSHASH_AMD_SK_SLOTS_PER_KMER=1.3 timeout 1500 python -m pytest tests/test_gpu_km_sweep.py tests/test_gpu_reference_data.py tests/test_gpu_streaming.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do
timeout 1200 python bench.py --workload c4 --no-cpu-baseline --no-file-query --no-other-paths --steps 10 --warmup 2 2>gpurun_out/r03_compact/c4_$i.err | tee gpurun_out/r03_compact/c4_$i.jsonl | python3 -c "
import json,sys; r=json.loads(sys.stdin.read()); print('c4', r['value']/1e9, r['ms_per_step'], r['config']['device_index_bytes'], r['config']['device_bytes_per_kmer'], r['other_mixes'])"
done
timeout 1500 python tools/perf_variants.py --k 63 --m 25 --bases 1500000000 --reads 1000000 2>gpurun_out/r03_compact/k63.err | tee gpurun_out/r03_compact/variants_k63.jsonl | cut -c1-200
