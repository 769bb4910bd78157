#!/bin/bash
# Heavy-key sweep: the C3 stand-in with every repeat family at 0.5x, 1x, 2x of the fitted recipe;
# throughput, bytes per k-mer, table histogram, resume share (kernel stats) for each.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r03_sweep
mkdir -p $OUT
export TMPDIR=/tmp
for s in 0.5 1.0 2.0; do
  python bench.py --repeat-scale $s --no-cpu-baseline --no-extra-mixes --no-other-paths --no-file-query --steps 10 --warmup 2 > $OUT/bench_$s.jsonl 2> $OUT/bench_$s.err
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$s -o t -- python bench.py --repeat-scale $s --no-cpu-baseline --no-extra-mixes --no-other-paths --no-file-query --steps 3 --warmup 1 > $OUT/trace_$s.log 2>&1
  find $OUT/trace_$s -name 't_kernel_stats.csv' -exec cp {} $OUT/kernel_stats_$s.csv \;
  rm -f /tmp/sshash_amd_bench_*.sshash
  python3 - $OUT $s <<'PY'
import json, sys, csv
out, s = sys.argv[1], sys.argv[2]
r = json.loads(open(f'{out}/bench_{s}.jsonl').read().strip().splitlines()[-1])
h = r['config']['table_histogram']
k = {}
for row in csv.DictReader(open(f'{out}/kernel_stats_{s}.csv')):
    for name in ('fast_lookup', 'resume_lookup', 'deferred_lookup'):
        if name in row['Name']:
            k[name] = round(float(row['AverageNs']) / 1e6, 3)
print(json.dumps({'repeat_scale': float(s), 'G_lookups_per_s': round(r['value'] / 1e9, 2), 'ms_per_step': r['ms_per_step'], 'roofline_frac': r['roofline']['frac'],
                  'device_bytes_per_kmer': r['config']['device_bytes_per_kmer'], 'kmers_under_heavy_keys_fraction': h['kmers_under_heavy_keys_fraction'],
                  'super_kmers_under_heavy_keys_fraction': h['super_kmers_under_heavy_keys_fraction'], 'kernel_ms_per_launch': k,
                  'positions_in_buckets_gt1_ratio_to_human': r['config']['index_statistics']['num_minimizer_positions_of_buckets_larger_than_1']['ratio'],
                  'kmers_in_skew_index_ratio_to_human': r['config']['index_statistics']['num_kmers_in_skew_index']['ratio']}))
PY
done | tee $OUT/sweep.jsonl
