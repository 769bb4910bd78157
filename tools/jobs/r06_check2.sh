#!/bin/bash
# round 6, second GPU check: the MPHF path with both strands resolved together against one after the other (same box, alternating),
# the density sweep of the table on C3, then the new full-size and eight-rank tests
cd "$(dirname "$0")/../.."
out=gpurun_out/${1:-r06_check2}; mkdir -p $out
export TMPDIR=/tmp SSHASH_BENCH_CACHE=/tmp
S="--steps 4 --warmup 1 --no-cpu-baseline --no-extra-mixes --no-file-query --no-other-workloads --no-line-probe --quiet-record"
for round in 1 2; do for hook in "" "mphf_strands_in_turn=1"; do
  name=paths_${round}_${hook:-together}
  SSHASH_AMD_TEST_HOOKS=$hook python bench.py $S --full-record $out/$name.json > $out/$name.jsonl 2>> $out/bench.err
  python3 - $out/$name.json "$name" <<'PY' | tee -a $out/paths.txt
import json, sys
r = json.load(open(sys.argv[1])); p = r["other_paths"]
print(sys.argv[2], "C3", round(r["value"] / 1e9, 2), "| " + " | ".join(f"{k} {v['lookups_per_s'] / 1e9:.2f} G/s" for k, v in p.items()))
PY
done; done
bash tools/jobs/r06_density_sweep.sh $(basename $out)/density c3
timeout 3000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_harness.py tests/test_gpu_baseline_workloads.py -m gpu -x -q --durations=12 > $out/pytest.txt 2>&1
tail -25 $out/pytest.txt
