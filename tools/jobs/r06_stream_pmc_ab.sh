#!/bin/bash
# round 6: what bounds the streaming kernel -- the shipped build against variants (turn order, waves per SIMD): time on two read sets, then the
# SQ counters of the run kernel for the two turn orders (one counter group per pass, kernel trace only)
cd "$(dirname "$0")/../.."
out=gpurun_out/${1:-r06_stream_pmc_ab}; mkdir -p $out
export TMPDIR=/tmp SSHASH_BENCH_CACHE=/tmp
for set in "c3 0.95" "c4 0.5"; do
  for lib in "" $(ls tools/ab/libvariant_*.so); do
    SSHASH_AMD_LIBRARY=${lib:+$PWD/$lib} python tools/debug/stream_ablation.py $set 2>> $out/err.txt | tee -a $out/ab.txt
  done
done
i=0
for g in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  for lib in "" tools/ab/libvariant_run_after_seed.so; do
    tag=$(basename ${lib:-shipped} .so)
    SSHASH_AMD_LIBRARY=${lib:+$PWD/$lib} timeout 900 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $out/pmc_$tag -o g$i -- python tools/debug/stream_ablation.py c3 0.95 > $out/pmc_${tag}_g$i.log 2>&1
  done
done
python3 - $out <<'PY' | tee $out/pmc_summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
for tag in ("shipped", "libvariant_run_after_seed"):
    res = collections.defaultdict(list)
    for f in sorted(glob.glob(f"{out}/pmc_{tag}/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            if "streaming_run_kernel" in r["Kernel_Name"]:
                res[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(tag, {c: f"{sum(v) / len(v):.4g}" for c, v in sorted(res.items())})
PY
rm -rf $out/pmc_shipped $out/pmc_libvariant_run_after_seed
