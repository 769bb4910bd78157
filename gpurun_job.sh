set -x
mkdir -p gpurun_out
cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null; python -c "import os; print(len(os.sched_getaffinity(0)))"
timeout 900 python -m pytest tests -m gpu -x -q --timeout 300 2>&1 | tail -5
for cfg in "8192 16777216 4 8" "8192 16777216 4 16" "8192 16777216 4 32" "8192 67108864 4 16" "8192 134217728 4 16" "8192 134217728 1 16" "1024 134217728 4 16" "128 134217728 4 16"; do ./tools/gather_bench $cfg; done
export TMPDIR=/tmp
ROOTDIR=$PWD
timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -3
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $ROOTDIR/gpurun_out/prof_trace --output-format csv -- python $ROOTDIR/bench.py --no-cpu-baseline --steps 5 --warmup 2 > $ROOTDIR/gpurun_out/prof_trace.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE -d $ROOTDIR/gpurun_out/prof_fetch --output-format csv -- python $ROOTDIR/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $ROOTDIR/gpurun_out/prof_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE -d $ROOTDIR/gpurun_out/prof_write --output-format csv -- python $ROOTDIR/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $ROOTDIR/gpurun_out/prof_write.log 2>&1
timeout 900 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $ROOTDIR/gpurun_out/prof_tcc --output-format csv -- python $ROOTDIR/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $ROOTDIR/gpurun_out/prof_tcc.log 2>&1
cd $ROOTDIR
find gpurun_out -name "*.csv" | head -30
tail -3 gpurun_out/prof_trace.log
for f in $(find gpurun_out/prof_trace -name "*kernel_stats.csv"); do head -8 $f; done
for d in prof_fetch prof_write prof_tcc; do for f in $(find gpurun_out/$d -name "*counter_collection.csv"); do python - "$f" <<'PY'
import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(list)
for r in rows:
    if 'lookup_kernel' in r.get('Kernel_Name',''):
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items(): print(sys.argv[1].split('/')[1], k, 'n=',len(v), 'mean=', sum(v)/len(v))
PY
done; done
# keep the merged output small: drop raw per-dispatch traces except stats
find gpurun_out -name "*kernel_trace.csv" -size +5M -delete
du -sh gpurun_out
