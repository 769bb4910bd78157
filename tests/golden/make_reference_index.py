#!/usr/bin/env python
"""Makes the fixtures of SURVEY.md 8(f1) -- a REFERENCE-built index and the reference's own answers over it -- on the day the reference's
lookup path can be compiled (it cannot in the container this repo was built in: /root/reference/external/pthash is an empty submodule
directory). Run where /root/reference is complete:

    make -C oracle ref-full                      # oracle/_ref/sshash, oracle/_ref/ref_lookup (plain g++ on the reference's own files)
    python tests/golden/make_reference_index.py  # this script

It writes, under tests/golden/ref_index/ (data: inputs and expected outputs, no reference source):
  se_k31_m13.sshash          `sshash build -i tests/golden/salmonella_enterica_k31_ust.fa.gz -k 31 -m 13` (tools/build.cpp:90-95: the
                             reference's own essentials::save of its dictionary -- the v5.1.1 byte format a loader has to read)
  se_k31_m13.canon.sshash    the same with --canonical
  queries.txt                20 000 ASCII k-mers: k-mers of the input in file order, every other one reverse-complemented, then k-mers
                             with one substituted base, then random ones
  lookups.jsonl              oracle/_ref/ref_lookup <index> lookup queries.txt: all eight lookup_result fields per query -- INCLUDING
                             minimizer_found of the misses, which this repo pins by its restatement alone until then
  lookups.canon.jsonl        the same over the canonical index
  query_report.json          ref_lookup <index> query tests/golden/SRR5833294.10K.fastq.gz: the six counters (the searches / extensions
                             split is the other thing pinned by the restatement alone)
  query_report.canon.json
  bench.json                 `sshash bench -i <index>` as it prints it (tools/perf.hpp:30-87): the genuine-reference CPU row on this host
tests/test_reference_index.py then stops skipping and holds the product (and the oracle) against these files."""
import gzip
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_BIN = os.path.join(ROOT, "oracle", "_ref")
OUT = os.path.join(HERE, "ref_index")
FASTA = os.path.join(HERE, "salmonella_enterica_k31_ust.fa.gz")
FASTQ = os.path.join(HERE, "SRR5833294.10K.fastq.gz")
K, M = 31, 13


def revcomp(s: str) -> str:
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


def make_queries(path: str) -> None:
    rng = np.random.default_rng(20260930)
    seqs = [l.strip() for l in gzip.open(FASTA, "rt") if l and l[0] != ">"]
    seqs = [s for s in seqs if len(s) >= K]
    out = []
    for s in seqs[:400]:  # positives in file order (ids 0, 1, 2, ...: test/check_from_file.hpp:66-72)
        for j in range(0, min(len(s) - K + 1, 25)):
            x = s[j:j + K]
            out.append(x if len(out) % 2 == 0 else revcomp(x))
    out = out[:10000]
    for x in list(out[:5000]):  # one substitution: absent, but sharing minimizers with the index
        p = int(rng.integers(0, K))
        c = "ACGT"[("ACGT".index(x[p]) + 1 + int(rng.integers(0, 3))) % 4]
        out.append(x[:p] + c + x[p + 1:])
    for _ in range(5000):
        out.append("".join("ACGT"[int(v)] for v in rng.integers(0, 4, K)))
    open(path, "w").write("\n".join(out) + "\n")


def main() -> int:
    sshash, ref_lookup = os.path.join(REF_BIN, "sshash"), os.path.join(REF_BIN, "ref_lookup")
    if not (os.path.exists(sshash) and os.path.exists(ref_lookup)):
        print("oracle/_ref/sshash or oracle/_ref/ref_lookup is missing: `make -C oracle ref-full` first (it needs the reference's "
              "external/pthash sources, absent from the checkout this repo was built against)", file=sys.stderr)
        return 1
    os.makedirs(OUT, exist_ok=True)
    queries = os.path.join(OUT, "queries.txt")
    make_queries(queries)
    for tag, extra in (("", []), (".canon", ["--canonical"])):
        index = os.path.join(OUT, f"se_k31_m13{tag}.sshash")
        subprocess.check_call([sshash, "build", "-i", FASTA, "-k", str(K), "-m", str(M), "-o", index, "-d", OUT] + extra)
        with open(os.path.join(OUT, f"lookups{tag}.jsonl"), "wb") as f:
            f.write(subprocess.check_output([ref_lookup, index, "lookup", queries]))
        report = subprocess.check_output([ref_lookup, index, "query", FASTQ]).decode().strip().splitlines()[-1]
        json.loads(report)
        open(os.path.join(OUT, f"query_report{tag}.json"), "w").write(report + "\n")
    bench = subprocess.run([sshash, "bench", "-i", os.path.join(OUT, "se_k31_m13.sshash")], capture_output=True, text=True, check=True)
    open(os.path.join(OUT, "bench.json"), "w").write(bench.stdout + bench.stderr)
    print("fixtures written to", OUT)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
