set -x
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q --timeout 600 2>&1 | tail -5
timeout 900 python tools/perf_variants.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/variants_regular.jsonl | tail -20
timeout 600 python bench.py --no-cpu-baseline --canonical 2>&1 | tail -1 | tee gpurun_out/bench_canonical.json
timeout 900 python bench.py --no-cpu-baseline --bases 2813192630 --mean-len 274 2>&1 | tail -4 | tee gpurun_out/bench_human.json
timeout 900 python bench.py --no-cpu-baseline --k 63 --m 25 --bases 1500000000 --mean-len 160 2>&1 | tail -3 | tee gpurun_out/bench_k63.json
export TMPDIR=/tmp; R=$PWD; cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/calib --output-format csv -- $R/tools/gather_bench 8192 16777216 4 16 3 > $R/gpurun_out/calib.log 2>&1
cd $R
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/calib/**/*counter_collection.csv', recursive=True):
    rows=[r for r in csv.DictReader(open(f)) if 'chase' in r['Kernel_Name']]
    for r in rows: print('calib', r['Counter_Name'], r['Counter_Value'], 'KB; reads=', 16777216*4, 'KB/64B-read expected', 16777216*4*64/1024)
PY
