#!/bin/bash
# Same-box A/B of one environment switch: alternating bench runs with VAR=A and VAR=B (the index is built once).
# usage: bash tools/jobs/r03_env_ab.sh <name> <VAR> <A> <B> <rounds> [bench args...]
set -u
cd "$(dirname "$0")/../.."
NAME=$1; VAR=$2; A=$3; B=$4; ROUNDS=$5; shift 5
OUT=gpurun_out/$NAME
mkdir -p $OUT
BENCH="python bench.py --no-cpu-baseline --no-extra-mixes --no-other-paths --no-file-query --steps 20 --warmup 3 $*"
for r in $(seq 1 $ROUNDS); do
  for v in $A $B; do
    env $VAR=$v $BENCH 2>> $OUT/bench.err | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); print('$VAR=$v', r['value']/1e9, r['ms_per_step'], r['roofline']['avg_kernel_ms'])" | tee -a $OUT/ab.txt
  done
done
