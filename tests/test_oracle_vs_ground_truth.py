"""CPU: the oracle (restated reference algorithm) against the input-order ground truth.

This is what pins the oracle's lookup logic without a runnable reference: ids are defined by input
order (reference test/check_from_file.hpp:66-72), so a table built straight from the FASTA is the
truth for every field of lookup_result.
"""
from __future__ import annotations

import numpy as np
import pytest

from conftest import ALL_SMALL_CASES

U64_FIELDS = ["kmer_id", "kmer_id_in_string", "kmer_offset", "string_id", "string_begin", "string_end"]


@pytest.mark.parametrize("case_name", ["case_se_regular", "case_se_canonical", "case_k63_regular"] + ALL_SMALL_CASES)
def test_oracle_matches_ground_truth(case_name, request):
    case = request.getfixturevalue(case_name)
    n = 50000 if case.gt.num_kmers > 100000 else 3000
    q = case.queries(n, n, seed=5)
    r = case.oracle.lookup_packed(q)
    g = case.gt.lookup(q)
    for f in U64_FIELDS:
        assert (r[f] == g[f]).all(), f
    found = g["found"]
    assert found.sum() >= n
    assert (r["kmer_orientation"][found] == g["kmer_orientation"][found]).all()
    # ids only, threaded variant
    assert (case.oracle.lookup_ids(q, num_threads=3) == g["kmer_id"]).all()


@pytest.mark.parametrize("case_name", ALL_SMALL_CASES)
def test_oracle_every_kmer_in_file_order(case_name, request):
    """check_from_file.hpp:38-83 on the oracle: every k-mer of the input, every other one
    reverse-complemented -> ids 0,1,2,... with the right orientation; access() round trip."""
    case = request.getfixturevalue(case_name)
    n = case.gt.num_kmers
    q = case.gt.kmers(np.arange(n)).reshape(n, case.W)
    q[::2] = case.gt._revcomp(q[::2].reshape(-1)).reshape(-1, case.W)
    r = case.oracle.lookup_packed(q.reshape(-1))
    assert (r["kmer_id"] == np.arange(n, dtype=np.uint64)).all()
    expect = np.ones(n, dtype=np.int64)
    expect[::2] = -1
    assert (r["kmer_orientation"] == expect).all()
    comp = str.maketrans("ACGT", "TGCA")
    for i in range(0, n, max(1, n // 300)):
        s_id = int(r["string_id"][i])
        pos = int(r["kmer_id_in_string"][i])
        truth = case.sequences[s_id][pos:pos + case.k]
        assert case.oracle.access(i) == truth
        assert case.dict.access(i) == truth  # host-side access of the product (used to draw positives)


def test_skew_cases_really_have_all_bucket_types(case_skew_regular, case_skew_canonical, case_small_k, case_k63_canonical):
    """The synthetic inputs must exercise SINGLETON, MIDLOAD and HEAVYLOAD buckets."""
    import re
    import subprocess
    import sys

    for case in (case_skew_regular, case_skew_canonical, case_small_k, case_k63_canonical):
        q = case.gt.kmers(np.arange(case.gt.num_kmers))
        per_query = [case.oracle.count_bytes(q[i * case.W:(i + 1) * case.W]) for i in range(0, case.gt.num_kmers, 7)]
        assert max(per_query) > min(per_query)  # different bucket classes cost different bytes


def test_algorithmic_bytes_rule(case_se_regular):
    """~100 B per lookup on a regular index for a 50/50 mix."""
    case = case_se_regular
    q = case.queries(5000, 5000, seed=1)
    per = case.oracle.count_bytes(q) / 10000
    assert 70 < per < 130


# ---- weights (reference test/check_from_file.hpp:229-275: weight(kmer_id) must equal the id-th abundance of the
# ---- input file, read in file order) -----------------------------------------------------------------------

def _file_weights(path, k):
    import gzip

    out = []
    with gzip.open(path, "rt") as f:
        for header in f:
            seq = next(f).strip()
            ln, ab = header.split(" ", 2)[1:]
            assert ln.startswith("LN:i:") and int(ln[5:]) == len(seq) and ab.startswith("ab:Z:")
            w = [int(x) for x in ab[5:].split()]
            assert len(w) == len(seq) - k + 1
            out.extend(w)
    return np.array(out, dtype=np.uint64)


@pytest.fixture(scope="module")
def weighted_case(tmp_path_factory):
    import sshash_amd
    from conftest import WEIGHTED_FASTA
    from oracle import oracle as O

    d = sshash_amd.Dictionary.build(WEIGHTED_FASTA, k=31, m=15, weighted=True, num_threads=4)
    p = str(tmp_path_factory.mktemp("w") / "weighted.sshash")
    d.save(p)
    return d, O.OracleIndex(p), _file_weights(WEIGHTED_FASTA, 31), p


def test_weights_equal_the_abundances_of_the_input_file(weighted_case):
    import sshash_amd

    d, oracle, want, path = weighted_case
    assert d.weighted() and d.num_kmers() == want.size
    ids = np.arange(want.size, dtype=np.uint64)
    assert (oracle.weights(ids) == want).all()          # the restatement against the file
    assert (d.weight(ids) == want).all()                # the host side of the library
    d2 = sshash_amd.Dictionary.load(path)               # round trip through the index file
    assert d2.weighted() and (d2.weight(ids[::97]) == want[::97]).all()
    with pytest.raises(sshash_amd.SSHashError):
        d.weight([want.size])                           # out of range


def test_unweighted_dictionary_refuses_weight_queries(case_skew_regular):
    import sshash_amd

    assert not case_skew_regular.dict.weighted()
    with pytest.raises(sshash_amd.SSHashError) as e:
        case_skew_regular.dict.weight([0])
    assert "does not store weights" in str(e.value)
    with pytest.raises(ValueError):
        case_skew_regular.oracle.weights([0])


def test_malformed_weight_headers_are_rejected(tmp_path):
    import sshash_amd

    seq = "ACGTTGCAAGGCTTAACCGGTTAAGGCCTTAACGT"  # 35 bases -> 5 k-mers at k = 31
    for header in (">0 LN:i:35 ab:Z:1 1 1", ">0 LN:i:34 ab:Z:1 1 1 1", ">0 ab:Z:1 1 1 1 1", ">0 LN:i:35 1 1 1 1 1"):
        p = tmp_path / "bad.fa"
        p.write_text(header + "\n" + seq + "\n")
        with pytest.raises(sshash_amd.SSHashError):
            sshash_amd.Dictionary.build(str(p), k=31, m=11, weighted=True)
    p = tmp_path / "good.fa"
    p.write_text(">0 LN:i:35 ab:Z:7 7 9 9 7\n" + seq + "\n")
    d = sshash_amd.Dictionary.build(str(p), k=31, m=11, weighted=True)
    assert list(d.weight(range(5))) == [7, 7, 9, 9, 7]


def test_bucket_statistics_against_an_independent_count(case_skew_regular, case_small_k):
    """A7 pinned from outside the builder: minimizers recomputed here with numpy straight from the sequences (the reference's rule:
    hash = (m-mer * 0x517cc1b727220a95) ^ magic, leftmost minimum, include/util.hpp:262-283, include/hash_util.hpp:91), a bucket =
    the DISTINCT positions of one minimizer (include/builder/util.hpp:61-78), classes by size (1 / 2..64 / > 64, partitions
    (64,128], (128,256] ...: src/builder/build_sparse_and_skew_index.cpp:141-147) -- against sshash_bucket_stats of the built index."""
    from oracle import oracle as O
    from oracle.ground_truth import encode_bases

    for case in (case_skew_regular, case_small_k):
        k, m = case.k, case.m
        magic = O.xxh64_u64(1, 0)
        positions = {}  # minimizer value -> set of absolute positions
        kmers_of = {}   # minimizer value -> number of k-mers
        base = 0
        mask = (1 << (2 * m)) - 1
        for s in case.sequences:
            codes = encode_bases(s).astype(object)
            mm = []
            v = 0
            for i, c in enumerate(codes):  # m-mers, first base in the low bits
                v |= int(c) << (2 * min(i, m - 1)) if i < m else 0
                if i >= m:
                    v = (v >> 2) | (int(c) << (2 * (m - 1)))
                if i >= m - 1:
                    mm.append(v & mask)
            hashes = [((x * 0x517CC1B727220A95) & ((1 << 64) - 1)) ^ magic for x in mm]
            w = k - m + 1
            for start in range(len(s) - k + 1):
                window = hashes[start:start + w]
                j = min(range(w), key=lambda t: (window[t], t))  # leftmost minimum
                mini = mm[start + j]
                positions.setdefault(mini, set()).add(base + start + j)
                kmers_of[mini] = kmers_of.get(mini, 0) + 1
            base += len(s)
        sizes = {mini: len(p) for mini, p in positions.items()}
        st = case.dict.bucket_stats()
        assert st["num_minimizers"] == len(sizes)
        assert st["num_minimizer_positions"] == sum(sizes.values())
        assert st["buckets_with_n_positions"] == [sum(1 for v in sizes.values() if v == n) for n in range(1, 17)]
        mid = [v for v in sizes.values() if 2 <= v <= 64]
        heavy = {mini: v for mini, v in sizes.items() if v > 64}
        assert st["num_buckets_larger_than_1_not_in_skew_index"] == len(mid) and st["num_minimizer_positions_of_buckets_larger_than_1"] == sum(mid)
        assert st["num_buckets_in_skew_index"] == len(heavy) and st["num_minimizer_positions_of_buckets_in_skew_index"] == sum(heavy.values())
        assert st["max_bucket_size"] == max(sizes.values())
        assert st["num_kmers_in_skew_index"] == sum(kmers_of[mini] for mini in heavy)
        if heavy:
            parts = [0] * len(st["num_kmers_in_skew_partition"])
            for mini, v in heavy.items():
                p = 0
                while p + 1 < len(parts) and v > (128 << p):
                    p += 1
                parts[p] += kmers_of[mini]
            assert st["num_kmers_in_skew_partition"] == parts
