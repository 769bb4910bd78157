"""GPU parity of the batched streaming query (six counters of streaming_query_report)."""
from __future__ import annotations

import gzip

import numpy as np
import pytest

from conftest import FASTQ, random_dna

pytestmark = pytest.mark.gpu


def _as_dict(rep):
    return {k: getattr(rep, k) for k in ("num_kmers", "num_positive_kmers", "num_negative_kmers", "num_invalid_kmers",
                                         "num_searches", "num_extensions")}


def _synthetic_reads(case, n_reads, seed, read_len=120):
    """Half of the reads are sampled from the indexed strings (either strand, 1% substitutions,
    'N' at rate 1e-2), half are random."""
    rng = np.random.default_rng(seed)
    comp = str.maketrans("ACGT", "TGCA")
    reads = []
    long_seqs = [s for s in case.sequences if len(s) >= case.k + 5]
    for i in range(n_reads):
        if i % 2 == 0:
            s = long_seqs[int(rng.integers(0, len(long_seqs)))]
            a = int(rng.integers(0, max(1, len(s) - case.k)))
            r = list(s[a:a + read_len])
            for j in range(len(r)):
                u = rng.random()
                if u < 0.01:
                    r[j] = "ACGT"[int(rng.integers(0, 4))]
                elif u < 0.02:
                    r[j] = "N"
            r = "".join(r)
            if rng.random() < 0.5:
                r = r.translate(comp)[::-1]
            if rng.random() < 0.3:
                r = r.lower()
        else:
            r = random_dna(rng, int(rng.integers(1, read_len)))
        reads.append(r)
    reads += ["", "A" * (case.k - 1), "N" * (case.k + 3)]
    return reads


@pytest.mark.parametrize("case_name", ["case_se_regular", "case_se_canonical", "case_skew_regular", "case_skew_canonical",
                                       "case_k63_canonical", "case_k63_regular", "case_small_k"])
def test_streaming_counters_match_oracle(case_name, request):
    case = request.getfixturevalue(case_name)
    d = case.dict.to_device(0)
    reads = _synthetic_reads(case, 3000, seed=17)
    got = _as_dict(d.streaming_query(reads))
    want = case.oracle.streaming_query(reads)
    assert got == want
    assert got["num_kmers"] == got["num_positive_kmers"] + got["num_negative_kmers"] + got["num_invalid_kmers"]
    assert got["num_positive_kmers"] == got["num_searches"] + got["num_extensions"]
    assert got["num_extensions"] > 0 and got["num_invalid_kmers"] > 0


def test_streaming_query_from_fastq_file(case_se_regular, case_se_canonical):
    """README.md:222-223 known answer num_kmers = 460000 for SRR5833294.10K at k=31; the other counters
    against the oracle, and num_positive against a brute-force set (test/check.cpp:61-98)."""
    with gzip.open(FASTQ, "rt") as f:
        lines = f.read().split("\n")
    reads = lines[1::4]
    reads = [r for r in reads if r]
    for case in (case_se_regular, case_se_canonical):
        d = case.dict.to_device(0)
        got = _as_dict(d.streaming_query_from_file(FASTQ))
        assert got["num_kmers"] == 460000
        assert got == case.oracle.streaming_query(reads)
        # independent count of positives
        pos = 0
        valid = set("ACGTacgt")
        kmers = []
        for r in reads:
            for i in range(len(r) - case.k + 1):
                x = r[i:i + case.k]
                if all(c in valid for c in x):
                    kmers.append(x)
        import sshash_amd

        arr = sshash_amd.encode_kmers(kmers, case.k)
        packed = np.zeros(len(kmers), dtype=np.uint64)
        codes = ((arr >> 1) & 3).astype(np.uint64)
        for j in range(case.k):
            packed |= codes[:, j] << np.uint64(2 * j)
        pos = int(case.gt.lookup(packed)["found"].sum())
        assert got["num_positive_kmers"] == pos


def test_multiline_fasta_and_unsupported_extension(case_skew_regular, tmp_path):
    case = case_skew_regular
    d = case.dict.to_device(0)
    s = case.sequences[0]
    p = tmp_path / "q.fa"
    # multiline: the header is NOT special-cased by the reference reader: its characters just make
    # k-mers invalid; an empty line ends a segment (src/query.cpp:9-47)
    p.write_text(">h\n" + s[:40] + "\n" + s[40:] + "\n\n>x\n" + s[:35] + "\n")
    got = _as_dict(d.streaming_query_from_file(str(p), multiline=True))
    want = case.oracle.streaming_query([">h" + s, ">x" + s[:35]])
    assert got == want
    # single-line FASTA reader: (header, sequence) line pairs -> (">h", s[:40]), (s[40:], ""), (">x", s[:35])
    single = _as_dict(d.streaming_query_from_file(str(p), multiline=False))
    assert single == case.oracle.streaming_query([s[:40], "", s[:35]])
    q = tmp_path / "q.txt"
    q.write_text("ACGT\n")
    rep = _as_dict(d.streaming_query_from_file(str(q)))
    assert rep == {k: 0 for k in rep}  # "unsupported query file format": empty report (src/query.cpp:169-171)
