#!/bin/bash
# does the physical chunk / fragment size move the random-line rate once lines are fetched cooperatively (the real ceiling)? round 2's sweep
# was made with one lane per line (19 G/s: bound elsewhere); hipMalloc against VMM chunks of 2 MiB ... 32 GiB, alternating
cd "$(dirname "$0")/../.."
out=gpurun_out/${NAME:-r04_tlb_coop}; mkdir -p $out
T=tools/tlb_probe; L=134217728
{ for round in 1 2; do
    timeout 300 $T 32768 64 malloc 0 0 $L 5 coop
    for c in "2 2" "64 64" "1024 1024" "32768 1024"; do timeout 300 $T 32768 64 vmm $c $L 5 coop; done
  done
  timeout 300 $T 40960 64 malloc 0 0 $L 5 coop
} 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    print(r['array_MiB'], r['mode'], r.get('chunk_MiB'), r.get('align_MiB'), 'va_align_log2', r.get('va_align_log2'), r['Greads_per_s'], r['ms_best'], r['ms_avg'])
" | tee $out/tlb_coop.txt
