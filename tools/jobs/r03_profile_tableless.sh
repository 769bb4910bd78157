#!/bin/bash
# Round 3: rocprofv3 kernel stats + PMC (HBM requests) of the two table-less paths on the C3 stand-in (10^8-query batch), and the
# VALU count of the k = 63 first pass. Summaries under gpurun_out/r03_tableless/.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r03_tableless
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_streaming.py -x -q -m gpu > $OUT/pytest_streaming.log 2>&1; tail -2 $OUT/pytest_streaming.log
COMMON="--no-cpu-baseline --no-extra-mixes --no-file-query --no-other-paths --steps 3 --warmup 1 --queries 100000000"
python bench.py $COMMON > $OUT/bench_table.jsonl 2> $OUT/bench_table.err   # builds and caches the index
for mode in directory mphf; do
  if [ $mode = directory ]; then export SSHASH_AMD_SKTABLE=0 SSHASH_AMD_DIRECTORY=1; else export SSHASH_AMD_SKTABLE=0 SSHASH_AMD_DIRECTORY=0; fi
  python bench.py $COMMON > $OUT/bench_$mode.jsonl 2> $OUT/bench_$mode.err
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$mode -o t -- python bench.py $COMMON > $OUT/trace_$mode.log 2>&1
  find $OUT/trace_$mode -name 't_kernel_stats.csv' -exec cp {} $OUT/kernel_stats_$mode.csv \;
  i=0
  for g in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum TCC_HIT_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum TCC_EA0_RDREQ_DRAM_sum"; do
    i=$((i+1))
    timeout 900 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $OUT/pmc_$mode -o g$i -- python bench.py $COMMON > $OUT/pmc_${mode}_g$i.log 2>&1
  done
done
unset SSHASH_AMD_SKTABLE SSHASH_AMD_DIRECTORY
K63="--workload c2 --k 63 --m 25 --bases 1500000000 --no-cpu-baseline --no-extra-mixes --no-other-paths --no-file-query --steps 3 --warmup 1"
python bench.py $K63 > $OUT/bench_k63.jsonl 2> $OUT/bench_k63.err
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/pmc_k63 -o g -- python bench.py $K63 > $OUT/pmc_k63.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, json, collections
out = sys.argv[1]
summary = {}
for mode in ('directory', 'mphf', 'k63'):
    res = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(f'{out}/pmc_{mode}/**/*counter_collection.csv', recursive=True)):
        for r in csv.DictReader(open(f)):
            n = r['Kernel_Name']
            if 'lookup_kernel' not in n: continue
            key = n.split('sshash_amd::')[1].split('<')[0]
            res[key][r['Counter_Name']].append(float(r['Counter_Value']))
    entry = {'per_launch_counter_averages': {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in res.items()}}
    if mode != 'k63':
        b = json.loads(open(f'{out}/bench_{mode}.jsonl').read().strip().splitlines()[-1])
        entry['bench'] = {'lookups_per_s': b['value'], 'ms_per_step': b['ms_per_step'], 'roofline': {k: b['roofline'][k] for k in ('achieved', 'frac', 'algorithmic_bytes_per_lookup', 'kernel')},
                          'device_index_bytes': b['config']['device_index_bytes']}
        stats = {}
        for r in csv.DictReader(open(f'{out}/kernel_stats_{mode}.csv')):
            if 'lookup_kernel' in r['Name']:
                stats[r['Name'].split('sshash_amd::')[1].split('<')[0]] = {'calls': int(r['Calls']), 'avg_ms': float(r['AverageNs']) / 1e6}
        entry['kernel_stats'] = stats
        pl = entry['per_launch_counter_averages']
        req = sum(d.get('TCC_EA0_RDREQ_sum', 0) + d.get('TCC_EA0_WRREQ_sum', 0) for d in pl.values())
        entry['hbm_requests_per_lookup'] = req / 1e8
    else:
        pl = entry['per_launch_counter_averages']
        for k, d in pl.items():
            if d.get('SQ_WAVES'): d['VALU_instructions_per_wave'] = d['SQ_INSTS_VALU'] / d['SQ_WAVES']
    summary[mode] = entry
json.dump(summary, open(out + '/summary.json', 'w'), indent=1)
for mode, e in summary.items():
    print(mode, json.dumps(e.get('bench', {})), json.dumps(e.get('kernel_stats', {})), e.get('hbm_requests_per_lookup'))
    for k, d in e['per_launch_counter_averages'].items():
        print('   ', k, {c: round(v) for c, v in sorted(d.items())})
PY
