#!/usr/bin/env python
"""Regenerates the golden vectors under tests/golden/ (run in the build container, where
/root/reference exists):

  ref_vectors.json   output of oracle/_ref/ref_vectors, a small generator compiled against the
                     reference's OWN include/kmer.hpp and external/cityhash (see oracle/ref_vectors.cpp
                     and oracle/Makefile) -- encodings, reverse complements, CityHash128WithSeed.
  xxh64_vectors.json XXH64 of one little-endian u64 for a few (value, seed) pairs from the python
                     `xxhash` package (the reference obtains its m-mer hash magic from
                     pthash::xxhash_64::hash(seed, 0), include/hash_util.hpp:88; PTHash itself is
                     absent from the reference checkout).
  Data files copied verbatim from the reference's data/ directory (inputs its own tools/tests use):
  salmonella_enterica_k31_ust.fa.gz, SRR5833294.10K.fastq.gz; se.ust.k63.head.fa.gz = the first 24
  records of se.ust.k63.fa.gz; salmonella_enterica.weighted.ust.k31.fa.gz =
  data/unitigs_stitched/with_weights/salmonella_enterica.ust.k31.fa.gz (headers carry the abundances).
  Round 3, copied verbatim as well: se.ust.k63.fa.gz (all 238 records: the reference's own k = 63 input),
  se.ust.k47.fa.gz (k = 47: two-word k-mers whose second word is half used), ecoli1_k31_ust.fa.gz,
  penicillium_chrysogenum_k31_ust.fa.gz (a eukaryote's repeat content), and data/queries/salmonella_enterica.fasta.gz -- a genome
  as MULTILINE FASTA, the input of the reference's multiline reader (src/query.cpp:9-47).
"""
import gzip
import json
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    out = subprocess.check_output([os.path.join(ROOT, "oracle", "_ref", "ref_vectors")])
    json.loads(out)
    open(os.path.join(HERE, "ref_vectors.json"), "wb").write(out)

    import xxhash

    vec = []
    for value in [0, 1, 2, 42, 1234567890, 2**63, 2**64 - 1, 0x0123456789ABCDEF]:
        for seed in [0, 1, 2**64 - 1]:
            vec.append({"value": value, "seed": seed, "xxh64": xxhash.xxh64(value.to_bytes(8, "little"), seed=seed).intdigest()})
    json.dump(vec, open(os.path.join(HERE, "xxh64_vectors.json"), "w"), indent=0)

    for rel in ["data/unitigs_stitched/salmonella_enterica_k31_ust.fa.gz", "data/queries/SRR5833294.10K.fastq.gz"]:
        dst = os.path.join(HERE, os.path.basename(rel))
        if not os.path.exists(dst):
            shutil.copy(os.path.join(REF, rel), dst)
    for rel in ["data/unitigs_stitched/se.ust.k63.fa.gz", "data/unitigs_stitched/se.ust.k47.fa.gz", "data/unitigs_stitched/ecoli1_k31_ust.fa.gz",
                "data/unitigs_stitched/penicillium_chrysogenum_k31_ust.fa.gz", "data/queries/salmonella_enterica.fasta.gz"]:
        dst = os.path.join(HERE, os.path.basename(rel))
        if not os.path.exists(dst):
            shutil.copy(os.path.join(REF, rel), dst)
    weighted = os.path.join(HERE, "salmonella_enterica.weighted.ust.k31.fa.gz")
    if not os.path.exists(weighted):
        shutil.copy(os.path.join(REF, "data/unitigs_stitched/with_weights/salmonella_enterica.ust.k31.fa.gz"), weighted)
    lines = gzip.open(os.path.join(REF, "data/unitigs_stitched/se.ust.k63.fa.gz"), "rb").read().split(b"\n")
    with gzip.open(os.path.join(HERE, "se.ust.k63.head.fa.gz"), "wb", compresslevel=9) as f:
        f.write(b"\n".join(lines[:48]) + b"\n")


if __name__ == "__main__":
    main()
