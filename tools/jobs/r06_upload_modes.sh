#!/bin/bash
cd "$(dirname "$0")/../.."
out=gpurun_out/${1:-r06_upload_modes}; mkdir -p $out
export SSHASH_BENCH_CACHE=/tmp
for p in 1 2 3 4; do python tools/debug/upload_modes.py 4 2>> $out/err.txt | tee -a $out/modes.txt; done
