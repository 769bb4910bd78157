#!/bin/bash
# round 5: C2 against the table's key length, same box, two alternating rounds (profiles/r05/c2_table_key_length_sweep.txt)
cd "$(dirname "$0")/../.."
out=gpurun_out/r05_c2_sweep; mkdir -p $out
S="--workload c2 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-mixes --no-other-paths --no-file-query --no-other-workloads --no-line-probe --quiet-record"
run() { name=$1; shift; envs=$1; shift
  env $envs python bench.py $S --full-record $out/$name.json > $out/$name.jsonl 2>> $out/bench.err
  python3 - $out/$name.json "$name" <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); st=r['config']['device_stats']
print(sys.argv[2], round(r['value']/1e9,2), 'G/s', r['ms_per_step'], 'ms', r['config']['device_bytes_per_kmer'], 'B/k-mer', 'heavy k-mers', st['sk_heavy_kmers'], 'keys', st['sk_keys'], 'load', st['sk_load_factor'])
PY
}
for round in 1 2; do
run m21_$round "A=1"
run m22_$round "SSHASH_AMD_SK_M=22"
run m23_$round "SSHASH_AMD_SK_M=23"
done
