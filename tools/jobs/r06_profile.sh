#!/bin/bash
# Round 6 profile of one bench workload (r05_profile.sh + the kernels' times from the per-dispatch trace, the bench's own short launches left out): the bench line (short), kernel trace, PMC passes (one counter group per run); summaries under gpurun_out/<name>/.
# usage: bash tools/jobs/r06_profile.sh <name> <bench args...>      (lookup or --streaming)
set -u
cd "$(dirname "$0")/../.."
NAME=${1:-r06_profile}; shift
OUT=gpurun_out/$NAME
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python bench.py --no-cpu-baseline --no-extra-mixes --no-file-query --no-other-paths --no-other-workloads --no-line-probe --quiet-record --steps 3 --warmup 1 $*"
$BENCH --full-record $OUT/bench_full.json > $OUT/bench.jsonl 2> $OUT/bench.err      # builds and caches the index
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $BENCH > $OUT/trace.log 2>&1
find $OUT/trace -name 't_kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
i=0
for g in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum TCC_HIT_sum" \
         "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum TCC_EA0_RDREQ_DRAM_sum" \
         "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCC_READ_REQ_sum" \
         "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES" \
         "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 1200 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $OUT/pmc -o g$i -- $BENCH > $OUT/pmc_g$i.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, sys, json, collections
out = sys.argv[1]
def key_of(n):
    if 'fast_lookup_kernel<' in n:
        args = n.split('fast_lookup_kernel<')[1].split('>(')[0].split(',')
        return 'fast' if args[-1].strip() == 'true' else 'fast_no_table'
    if 'resume_lookup' in n: return 'resume'
    if 'scan_lookup' in n: return 'scan'
    if 'deferred' in n: return 'deferred'
    if 'streaming_run' in n: return 'streaming'
    if 'streaming_kernel' in n: return 'streaming'
    if 'stream_pack' in n: return 'stream_pack'
    if 'stream_encode' in n: return 'stream_encode'
    if 'stream_classify' in n: return 'stream_classify'
    if 'stream_tile' in n: return 'stream_tile_reads'
    if 'lookup_kernel' in n: return 'generic'
    return None
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + '/pmc/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        k = key_of(r['Kernel_Name'])
        if k: res[k][r['Counter_Name']].append(float(r['Counter_Value']))
# (the streaming bench also runs one short launch for the oracle check: per-launch averages are over the long ones only when the short one is dropped)
summary = {}
for k, d in res.items():
    summary[k] = {}
    for c, v in d.items():
        big = [x for x in v if x >= 0.5 * max(v)] if k in ('streaming', 'stream_pack') else v
        summary[k][c] = sum(big) / len(big)
stats = {}
for r in csv.DictReader(open(out + '/kernel_stats.csv')):
    k = key_of(r['Name'])
    if k: stats[k] = {'calls': int(r['Calls']), 'avg_ns_all_calls': float(r['AverageNs']), 'max_ns': float(r.get('MaxNs', 0) or 0), 'name': r['Name'][:120]}
# rocprofv3's average is over EVERY launch of a kernel -- for the streaming bench that includes the oracle check's launch over the first few
# thousand reads (0.7 ms beside 11), which made round 5's "average" 20 % shorter than any timed launch ("a fifth of the step is
# outside the kernels"). avg_ns = the launches of the timed region: those at least half as long as the longest, from the per-dispatch trace.
dur = collections.defaultdict(list)
for f in glob.glob(out + '/trace/**/t_kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = key_of(r['Kernel_Name'])
        if k: dur[k].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in dur.items():
    big = [x for x in v if x >= 0.5 * max(v)]
    stats.setdefault(k, {})['avg_ns'] = sum(big) / len(big)
    stats[k]['timed_launches'] = len(big)
for k in stats: stats[k].setdefault('avg_ns', stats[k].get('avg_ns_all_calls'))
json.dump({'per_launch_counter_averages': summary, 'kernel_stats': stats}, open(out + '/summary.json', 'w'), indent=1)
print(json.dumps(stats, indent=1))
for k, d in summary.items():
    print(k, {c: round(v) for c, v in sorted(d.items())})
PY
rm -rf $OUT/trace $OUT/pmc     # (tens of megabytes of per-dispatch rows: what is kept is summary.json, kernel_stats.csv, bench.jsonl)
tail -1 $OUT/bench.jsonl | cut -c1-300
