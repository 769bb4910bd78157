#!/bin/bash
# round 6: footprint against rate on ONE box -- C3 (and C2, C4 with "all") with the super-k-mer table packed at 2.5 ... 1.25 slots per item
# (keys' region; the heavy keys' k-mers' region scaled along). One line per point: B/k-mer of the replica, G lookups/s, load factor, unplaced.
cd "$(dirname "$0")/../.."
out=gpurun_out/${1:-r06_density_sweep}; mkdir -p $out
export TMPDIR=/tmp SSHASH_BENCH_CACHE=/tmp
S="--steps 6 --warmup 2 --no-cpu-baseline --no-extra-mixes --no-file-query --no-other-paths --no-other-workloads --no-line-probe --quiet-record"
for w in ${2:-c3}; do
for point in "2.5 1.75" "2.0 1.5" "1.6 1.35" "1.4 1.25" "1.25 1.2"; do set -- $point
  [ $w = c4 ] && [ $1 = 2.5 ] && set -- 3.0 2.5     # (k <= 63: the default there)
  SSHASH_AMD_TEST_HOOKS="slots_per_key=$1,slots_per_kmer=$2" python bench.py $S --workload $w --full-record $out/${w}_$1.json > $out/${w}_$1.jsonl 2>> $out/bench.err
  python3 - $out/${w}_$1.json $w $1 $2 <<'PY' | tee -a $out/sweep.txt
import json, sys
r = json.load(open(sys.argv[1])); c = r["config"]; st = c["device_stats"]
print(f"{sys.argv[2]} slots_per_item {sys.argv[3]} slots_per_heavy_kmer {sys.argv[4]}: {c['device_bytes_per_kmer']} B/k-mer ({st['bytes'] / 1e9:.2f} GB; table {st['sk_bytes'] / 1e9:.2f} GB, load factor {st['sk_load_factor']}), "
      f"{r['value'] / 1e9:.2f} G lookups/s ({r['ms_per_step']} ms/step), deferred keys {st['sk_deferred_keys']}, parity: ids equal oracle on {c['ids_equal_oracle_on_queries']}")
PY
done; done
