#!/bin/bash
# round 6: the streaming kernel against the table's key length (SSHASH_AMD_SK_M): shorter keys = longer super-k-mers = fewer seeds per
# substitution, but more k-mers under heavy keys (round 4 measured this on the per-base kernel: -10 %; the event kernel is another machine)
cd "$(dirname "$0")/../.."
out=gpurun_out/${1:-r06_stream_key_length}; mkdir -p $out
export TMPDIR=/tmp SSHASH_BENCH_CACHE=/tmp
for m in 21 19 17 23; do SSHASH_AMD_SK_M=$m python tools/debug/stream_ablation.py c3 0.95 2>> $out/err.txt | sed "s/^/table key $m: /" | tee -a $out/key_length.txt; done
for m in 31 29 27; do SSHASH_AMD_SK_M=$m python tools/debug/stream_ablation.py c4 0.5 2>> $out/err.txt | sed "s/^/table key $m: /" | tee -a $out/key_length.txt; done
