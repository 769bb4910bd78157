#!/bin/bash
# Round 6: where does the streaming step's time go outside its two kernels? One rocprofv3 run with the kernel, memory-copy and HIP API
# traces of `bench.py --streaming` (no counters in it: gpurun refuses --pmc together with trace domains), summarised per step.
# usage: bash tools/jobs/r06_stream_trace.sh <name> <bench args...>
set -u
cd "$(dirname "$0")/../.."
NAME=${1:-r06_stream_trace}; shift
OUT=gpurun_out/$NAME
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python bench.py --no-cpu-baseline --no-extra-mixes --no-file-query --no-other-paths --no-other-workloads --no-line-probe --quiet-record --steps 6 --warmup 2 $*"
$BENCH --full-record $OUT/bench_full.json > $OUT/bench.jsonl 2> $OUT/bench.err      # builds and caches the index
timeout 1200 rocprofv3 --kernel-trace --memory-copy-trace --hip-trace --stats --output-format csv -d $OUT/trace -o t -- $BENCH > $OUT/trace.log 2>&1
for f in kernel_stats hip_api_stats memory_copy_stats; do find $OUT/trace -name "t_$f.csv" -exec cp {} $OUT/$f.csv \; ; done
python3 - $OUT <<'PY'
import csv, glob, sys, json
out = sys.argv[1]
def rows(pat):
    fs = glob.glob(out + '/trace/**/' + pat, recursive=True)
    return list(csv.DictReader(open(fs[0]))) if fs else []
K = rows('t_kernel_trace.csv'); A = rows('t_hip_api_trace.csv'); M = rows('t_memory_copy_trace.csv')
def short(n): return n.split('(')[0].split('<')[0].replace('void ', '').replace('sshash_amd::', '').replace('(anonymous namespace)::', '')
runs = [r for r in K if 'streaming_run_kernel' in r['Kernel_Name']]
big = max(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in runs)
runs = [r for r in runs if int(r['End_Timestamp']) - int(r['Start_Timestamp']) > big / 2]     # (the oracle check's short launch dropped)
ev = []
for r in K: ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K ' + short(r['Kernel_Name'])))
for r in M: ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'M ' + r.get('Direction', r.get('Name', 'copy'))))
for r in A: ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'A ' + r['Function']))
ev.sort()
lines = []
for i in range(1, len(runs)):
    a, b = int(runs[i - 1]['End_Timestamp']), int(runs[i]['End_Timestamp'])
    lines.append(f"--- step ending with run kernel #{i}: {(b - a) / 1e6:.3f} ms between the ends of two consecutive streaming_run_kernel launches")
    for s, e, n in ev:
        if s >= a and s < b and (n[0] != 'A' or e - s > 20000 or 'Launch' in n or 'Malloc' in n or 'Free' in n or 'Sync' in n or 'Memcpy' in n):
            lines.append(f"   +{(s - a) / 1e6:8.3f} ms  {(e - s) / 1e6:8.3f} ms  {n}")
open(out + '/step_timeline.txt', 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines[: 3 * (len(lines) // max(1, len(runs) - 1)) + 3]))
PY
rm -rf $OUT/trace
tail -1 $OUT/bench.jsonl | cut -c1-300
