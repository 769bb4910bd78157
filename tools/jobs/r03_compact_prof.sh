#!/bin/bash
# kernel stats of the C4 bench with the previous build and with this one (same box)
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r03_compact_prof; mkdir -p $OUT; export TMPDIR=/tmp
B="python bench.py --workload c4 --no-cpu-baseline --no-extra-mixes --no-file-query --no-other-paths --steps 3 --warmup 1"
$B > /dev/null 2>&1
cp sshash_amd/libsshash_amd.so /tmp/new.so
for v in old new; do
  if [ $v = old ]; then cp tools/debug/libsshash_amd_old.so sshash_amd/libsshash_amd.so; else cp /tmp/new.so sshash_amd/libsshash_amd.so; fi
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$v -o t -- $B > $OUT/trace_$v.log 2>&1
  find $OUT/trace_$v -name 't_kernel_stats.csv' -exec cp {} $OUT/kernel_stats_$v.csv \;
  echo $v; head -6 $OUT/kernel_stats_$v.csv | cut -c1-60,150-260
done
cp /tmp/new.so sshash_amd/libsshash_amd.so
