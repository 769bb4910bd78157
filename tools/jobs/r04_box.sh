#!/bin/bash
# what differs between the pool's boxes? (the headline ranges 36.5-41.2 G lookups/s over them; the random-line probe does not): sustained
# vector-ALU clock (tools/alu_bench: fixed work, ms), the random-line probe, clocks and power as rocm-smi reports them while the C3 bench runs
cd "$(dirname "$0")/../.."
out=gpurun_out/${NAME:-r04_box}; mkdir -p $out
{
echo "== host $(hostname) $(date -u +%H:%M:%S)"
rocm-smi --showproductname --showmaxpower --showperflevel --showclocks 2>/dev/null | grep -v "^=\|^$" | head -30
echo "== alu_bench (fixed work: ms ~ 1 / sustained clock)"; tools/alu_bench | head -3
echo "== random-line probe"; tools/tlb_probe 32768 64 malloc 0 0 134217728 5 coop | cut -c1-60,280-420
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|fclk\|power" | tr '\n' ' '; echo; sleep 1; done ) > $out/smi_during_bench.txt &
sampler=$!
echo "== c3 bench"; python bench.py --no-cpu-baseline --no-extra-mixes --no-file-query --no-other-workloads --no-other-paths --no-line-probe --steps 300 --warmup 5 --workload c3 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']/1e9,2), r['ms_per_step'])"
kill $sampler
echo "== rocm-smi while the bench ran (middle samples)"; grep -o "sclk clock level: [0-9S]*: ([0-9]*Mhz)\|Power (W): [0-9.]*" $out/smi_during_bench.txt | paste - - | sort | uniq -c | sort -rn | head -8
echo "== alu_bench again"; tools/alu_bench | head -1
} 2>&1 | tee $out/box.txt
