#!/bin/bash
# Round 2, experiment 3: is the random-read ceiling per 64-byte request or per DRAM row activation? Units of 128 and 256
# bytes read by 8 / 16 adjacent lanes in one load instruction (one translation, 2 / 4 adjacent 64-byte requests).
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r02_tlb3
mkdir -p $OUT
T=tools/tlb_probe
L=134217728
{
for w in 64 128 256; do
  timeout 300 $T 32768 $w malloc 0 0 $L 5 coop
done
} > $OUT/tlb_probe.jsonl 2> $OUT/tlb_probe.err
cut -c1-60,300-420 $OUT/tlb_probe.jsonl
head -5 $OUT/tlb_probe.err
