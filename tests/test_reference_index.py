"""the reference's OWN index and the reference's OWN answers, the day they can be made. tests/golden/ref_index/ is
written by tests/golden/make_reference_index.py from binaries `make -C oracle ref-full` compiles out of /root/reference -- which needs
the third-party sources of the reference's lookup path (external/pthash and what it nests), an empty submodule directory in the
checkout this repo was built against. Until those fixtures exist every test here SKIPS and says why; with them:

  * the CPU oracle (restatement) over an index this repo builds from the same FASTA returns the reference's ids, orientations and
    string fields for every query -- and the reference's `minimizer_found` for the misses, the field DESIGN.md section 2 lists as
    pinned by the restatement alone;
  * the six streaming counters, incl. the searches / extensions split, equal the reference's on its own FASTQ;
  * (gpu) the product returns the same through the C ABI;
  * a loader for the reference's byte format (`sshash_load_reference`, not written: it cannot be tested before these files exist)
    would be held against se_k31_m13.sshash here."""
from __future__ import annotations

import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

REF = os.path.join(GOLDEN, "ref_index")
WHY = ("tests/golden/ref_index/ is absent: the reference's lookup path cannot be compiled in the build container (external/pthash is an "
       "empty submodule) -- run `make -C oracle ref-full && python tests/golden/make_reference_index.py` where it can")
needs_fixtures = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "lookups.jsonl")), reason=WHY)
FASTA = os.path.join(GOLDEN, "salmonella_enterica_k31_ust.fa.gz")
FASTQ = os.path.join(GOLDEN, "SRR5833294.10K.fastq.gz")
FIELDS = ("kmer_id", "kmer_id_in_string", "kmer_offset", "string_id", "string_begin", "string_end")
COUNTERS = ("num_kmers", "num_positive_kmers", "num_negative_kmers", "num_invalid_kmers", "num_searches", "num_extensions")


def reference_rows(tag):
    return [json.loads(l) for l in open(os.path.join(REF, f"lookups{tag}.jsonl")) if l.strip()]


def pack(kmers):
    code = {"A": 0, "C": 1, "T": 2, "G": 3}  # include/kmer.hpp:118
    return np.array([sum(code[c] << (2 * i) for i, c in enumerate(x)) for x in kmers], dtype=np.uint64)


@needs_fixtures
@pytest.mark.parametrize("tag,canonical", [("", False), (".canon", True)])
def test_oracle_equals_the_reference_on_its_own_index(tag, canonical, tmp_path):
    import sshash_amd
    from oracle import oracle as O

    rows = reference_rows(tag)
    d = sshash_amd.Dictionary.build(FASTA, k=31, m=13, canonical=canonical, num_threads=0)
    path = str(tmp_path / "own.sshash")
    d.save(path)
    got = O.OracleIndex(path).lookup_packed(pack([r["kmer"] for r in rows]))
    for f in FIELDS:
        assert (got[f] == np.array([r[f] for r in rows], dtype=np.uint64)).all(), f
    assert (got["kmer_orientation"] == np.array([r["kmer_orientation"] for r in rows], dtype=np.int64)).all()
    assert (got["minimizer_found"].astype(bool) == np.array([r["minimizer_found"] for r in rows])).all(), "minimizer_found (misses included)"


@needs_fixtures
@pytest.mark.parametrize("tag,canonical", [("", False), (".canon", True)])
def test_oracle_streaming_counters_equal_the_reference(tag, canonical, tmp_path):
    import gzip

    import sshash_amd
    from oracle import oracle as O

    want = json.load(open(os.path.join(REF, f"query_report{tag}.json")))
    d = sshash_amd.Dictionary.build(FASTA, k=31, m=13, canonical=canonical, num_threads=0)
    path = str(tmp_path / "own.sshash")
    d.save(path)
    reads = [l.strip().encode() for i, l in enumerate(gzip.open(FASTQ, "rt")) if i % 4 == 1]
    got = O.OracleIndex(path).streaming_query(reads)
    assert {f: int(got[f]) for f in COUNTERS} == {f: int(want[f]) for f in COUNTERS}


@needs_fixtures
@pytest.mark.gpu
@pytest.mark.parametrize("tag,canonical", [("", False), (".canon", True)])
def test_product_equals_the_reference_on_its_own_index(tag, canonical):
    import sshash_amd

    rows = reference_rows(tag)
    d = sshash_amd.Dictionary.build(FASTA, k=31, m=13, canonical=canonical, num_threads=0)
    d.to_device(0)
    got = d.lookup(pack([r["kmer"] for r in rows]), full=True)
    for f in FIELDS:
        assert (getattr(got, f) == np.array([r[f] for r in rows], dtype=np.uint64)).all(), f
    assert (got.kmer_orientation == np.array([r["kmer_orientation"] for r in rows], dtype=np.int8)).all()
    assert (got.minimizer_found.astype(bool) == np.array([r["minimizer_found"] for r in rows])).all()
    want = json.load(open(os.path.join(REF, f"query_report{tag}.json")))
    rep = d.streaming_query_from_file(FASTQ)
    assert {f: int(getattr(rep, f)) for f in COUNTERS} == {f: int(want[f]) for f in COUNTERS}


def test_the_recipe_is_in_place():
    """what has to exist for the fixtures to be made: the Makefile target, the driver, the script -- and, here, the reason they have not run"""
    from conftest import ROOT

    mk = open(os.path.join(ROOT, "oracle", "Makefile")).read()
    assert "ref-full:" in mk and "tools/sshash.cpp" in mk and "ref_lookup.cpp" in mk
    assert os.path.exists(os.path.join(ROOT, "oracle", "ref_lookup.cpp")) and os.path.exists(os.path.join(GOLDEN, "make_reference_index.py"))
    if os.path.isdir("/root/reference/external/pthash"):
        if not os.path.exists("/root/reference/external/pthash/include/pthash.hpp"):
            assert not os.path.exists(os.path.join(REF, "lookups.jsonl")), "fixtures exist although the reference cannot be built here?"
