"""CPU: the oracle's streaming state machine against its own point lookups (the reference asserts this
equality at include/streaming_query.hpp:107), and the synthetic benchmark generator."""
from __future__ import annotations

import numpy as np
import pytest

from conftest import random_dna


@pytest.mark.parametrize("case_name", ["case_skew_regular", "case_skew_canonical", "case_k63_canonical", "case_small_k"])
def test_streaming_results_equal_point_lookups(case_name, request):
    case = request.getfixturevalue(case_name)
    rng = np.random.default_rng(4)
    comp = str.maketrans("ACGT", "TGCA")
    total = {"pos": 0, "neg": 0, "inv": 0}
    for t in range(60):
        s = case.sequences[int(rng.integers(0, len(case.sequences)))]
        read = s
        if t % 3 == 0:
            read = s.translate(comp)[::-1]
        if t % 4 == 1 and len(read) > case.k + 4:
            i = int(rng.integers(0, len(read)))
            read = read[:i] + "N" + read[i + 1:]
        if t % 5 == 2:
            read = read + random_dna(rng, case.k + 7)
        res = case.oracle.streaming_read(read)
        valid = set("ACGTacgt")
        for i in range(len(read) - case.k + 1):
            kmer = read[i:i + case.k]
            if not all(c in valid for c in kmer):
                assert res["kmer_id"][i] == np.uint64(0xFFFFFFFFFFFFFFFF)
                total["inv"] += 1
                continue
            point = case.oracle.lookup_ascii(np.frombuffer(kmer.encode(), dtype=np.uint8))[0]
            # equal_lookup_result, include/util.hpp:107-141
            for f in ("kmer_id", "kmer_id_in_string", "string_id", "string_begin", "string_end"):
                assert res[f][i] == point[f], (case_name, t, i, f)
            if point["kmer_id"] != np.uint64(0xFFFFFFFFFFFFFFFF):
                assert res["kmer_orientation"][i] == point["kmer_orientation"]
                total["pos"] += 1
            else:
                total["neg"] += 1
    assert total["pos"] > 100 and total["inv"] > 0


def test_streaming_counters_identities(case_skew_regular):
    case = case_skew_regular
    rng = np.random.default_rng(8)
    reads = [case.sequences[i % len(case.sequences)] for i in range(40)] + [random_dna(rng, 90) for _ in range(40)] + ["", "ACG", "N" * 50]
    rep = case.oracle.streaming_query(reads)
    assert rep["num_kmers"] == sum(max(0, len(r) - case.k + 1) for r in reads)  # src/query.cpp:93-94
    assert rep["num_kmers"] == rep["num_positive_kmers"] + rep["num_negative_kmers"] + rep["num_invalid_kmers"]
    assert rep["num_positive_kmers"] == rep["num_searches"] + rep["num_extensions"]  # streaming_query.hpp:113
    assert rep["num_extensions"] > rep["num_searches"] > 0  # whole strings: almost everything extends


def test_synthetic_spss_has_no_duplicate_kmers_and_all_bucket_classes():
    import sshash_amd
    from oracle.ground_truth import _revcomp_u64
    from sshash_amd.synthetic import draw_queries, make_spss

    k = 31
    words, ends = make_spss(3_000_000, k=k, m=15, num_motifs=60, seed=5)
    codes = ((words[:, None] >> (np.arange(32, dtype=np.uint64) * np.uint64(2))) & np.uint64(3)).reshape(-1)[: int(ends[-1])]
    n = codes.size - k + 1
    km = np.zeros(n, dtype=np.uint64)
    for j in range(k):
        km |= codes[j:j + n] << np.uint64(2 * j)
    inner = ends[1:-1].astype(np.int64)
    starts = np.zeros(codes.size + 2, dtype=np.int32)
    np.add.at(starts, np.maximum(inner - k + 1, 0), 1)
    np.add.at(starts, inner, -1)
    km = km[np.cumsum(starts)[:n] == 0]
    canon = np.minimum(km, _revcomp_u64(km, k))
    assert np.unique(canon).size == km.size  # a spectrum-preserving string set: every canonical k-mer once
    d = sshash_amd.Dictionary.build_from_packed(words, ends, k=k, m=15, num_threads=4)
    assert d.num_kmers() == km.size
    q = draw_queries(d, 10000, 0.5, seed=1)
    assert q.size == 10000 and q.dtype == np.uint64
    # positives really are k-mers of the set (either strand)
    qc = np.minimum(q, _revcomp_u64(q, k))
    assert 4900 <= int(np.isin(qc, canon).sum()) <= 5100


def test_repeat_family_spss_is_a_set_and_reports_the_builders_statistics():
    """sshash_amd/repeats.py: diverged families + cores, de-duplicated -- every canonical k-mer once -- and
    sshash_bucket_stats agrees with a histogram computed here from the strings alone."""
    import sshash_amd
    from oracle.ground_truth import _revcomp_u64
    from sshash_amd.repeats import make_repeat_spss, statistics_vs_target

    k, m = 31, 21
    classes = [{"copies": 2, "length": 300, "divergence": 0.02, "families": 300},
               {"copies": 40, "length": 150, "divergence": 0.1, "families": 10},
               {"copies": 300, "length": 120, "core": 23, "families": 3.5},
               {"copies": 200, "length": 120, "core": 21, "families": 1.8}]  # one whole family + a fractional one: distinct flanks
    words, ends = make_repeat_spss(1_500_000, k=k, classes=classes, reference_bases=1_500_000, seed=11, device="cpu")
    w2, e2 = make_repeat_spss(1_500_000, k=k, classes=classes, reference_bases=1_500_000, seed=11, device="cpu")
    assert (words == w2).all() and (ends == e2).all()  # deterministic
    codes = ((words[:, None] >> (np.arange(32, dtype=np.uint64) * np.uint64(2))) & np.uint64(3)).reshape(-1)[: int(ends[-1])]
    n = codes.size - k + 1
    km = np.zeros(n, dtype=np.uint64)
    for j in range(k):
        km |= codes[j:j + n] << np.uint64(2 * j)
    inner = ends[1:-1].astype(np.int64)
    crossing = np.zeros(codes.size + 2, dtype=np.int32)
    np.add.at(crossing, np.maximum(inner - k + 1, 0), 1)
    np.add.at(crossing, inner, -1)
    km = km[np.cumsum(crossing)[:n] == 0]
    canon = np.minimum(km, _revcomp_u64(km, k))
    assert np.unique(canon).size == km.size  # a spectrum-preserving string set
    assert int((np.diff(ends.astype(np.int64)) < k).sum()) == 0
    d = sshash_amd.Dictionary.build_from_packed(words, ends, k=k, m=m, num_threads=4)
    s = d.bucket_stats()
    assert s["num_kmers"] == km.size and s["num_strings"] == ends.size - 1 and s["num_bases"] == int(ends[-1])
    assert s["num_minimizers"] == d.num_minimizers()
    assert s["num_minimizers"] == s["buckets_with_n_positions"][0] + s["num_buckets_larger_than_1_not_in_skew_index"] + s["num_buckets_in_skew_index"]
    assert s["num_buckets_in_skew_index"] >= 1 and s["max_bucket_size"] > 64  # the cores of 300 copies
    assert s["buckets_with_n_positions"][1] > 100  # the pairs' shared m-mers around their substitutions
    assert s["num_minimizer_positions"] == s["buckets_with_n_positions"][0] + s["num_minimizer_positions_of_buckets_larger_than_1"] + \
        s["num_minimizer_positions_of_buckets_in_skew_index"]
    assert sum(s["num_kmers_in_skew_partition"]) == s["num_kmers_in_skew_index"]
    cmp = statistics_vs_target(s, "human_k31")  # the shipped recipe's targets load and scale
    assert cmp["num_kmers"]["achieved"] == km.size and 0 < cmp["scale"] < 1e-2


def test_repeat_family_spss_k63_is_a_set():
    """The same generator at k = 63 (two-word canonical forms): every canonical 63-mer once, strings >= k,
    cores shorter than k shared by hundreds of strings give a skew-index bucket at m = 25."""
    import sshash_amd
    from sshash_amd.repeats import make_repeat_spss, statistics_vs_target

    k, m = 63, 25
    classes = [{"copies": 2, "length": 400, "divergence": 0.02, "families": 200},
               {"copies": 30, "length": 300, "divergence": 0.1, "families": 6},
               {"copies": 200, "length": 200, "core": 45, "families": 2.5}]
    words, ends = make_repeat_spss(800_000, k=k, classes=classes, reference_bases=800_000, seed=5, device="cpu")
    codes = ((words[:, None] >> (np.arange(32, dtype=np.uint64) * np.uint64(2))) & np.uint64(3)).reshape(-1)[: int(ends[-1])]
    text = "".join("ACGT"[c] for c in codes.tolist())
    comp = str.maketrans("ACGT", "TGCA")
    seen = set()
    total = 0
    e = ends.astype(np.int64)
    for b, t in zip(e[:-1].tolist(), e[1:].tolist()):
        assert t - b >= k
        for i in range(b, t - k + 1):
            s = text[i:i + k]
            r = s.translate(comp)[::-1]
            seen.add(s if s < r else r)
            total += 1
    assert len(seen) == total  # a spectrum-preserving string set
    d = sshash_amd.Dictionary.build_from_packed(words, ends, k=k, m=m, num_threads=4)
    s = d.bucket_stats()
    assert s["num_kmers"] == total and s["num_buckets_in_skew_index"] >= 1 and s["max_bucket_size"] > 64
    cmp = statistics_vs_target(s, "human_k63")
    assert cmp["num_kmers"]["achieved"] == total and 0 < cmp["scale"] < 1e-2


def test_the_stand_in_generator_cuts_chance_duplicates_out():
    """A spectrum-preserving string set holds every k-mer once. The stand-in generator de-duplicates family by family; k-mers of
    different families coincide by chance once or twice per 10^9 (round 5: the full-size C2 stand-in held one, and
    lookup(access(id)) == id failed for that id). _duplicate_kmer_starts / _cut_kmers_out (sshash_amd/repeats.py) find and remove them:
    here on strings with a planted copy and a planted reverse complement -- every other k-mer stays, none occurs twice."""
    import torch

    from sshash_amd import repeats as R

    k = 31
    g = torch.Generator()
    g.manual_seed(3)
    codes = torch.randint(0, 4, (6000,), generator=g, dtype=torch.uint8)
    lens = torch.tensor([1000, 1500, 2500, 1000])
    codes[3200:3231] = codes[100:131]                  # a k-mer of string 0 once more in string 2
    codes[1400:1431] = codes[100:131].flip(0) ^ 2      # and its reverse complement in string 1
    codes[5990:6000] = codes[40:50]                    # (ten bases: no k-mer)

    def kmer_multiset(c, l):
        out, a = {}, 0
        for n in l.tolist():
            s = c[a:a + n].numpy()
            for i in range(n - k + 1):
                x = s[i:i + k]
                key = min(bytes(x), bytes(x[::-1] ^ 2))
                out[key] = out.get(key, 0) + 1
            a += n
        return out

    before = kmer_multiset(codes, lens)
    assert max(before.values()) == 3
    starts = R._duplicate_kmer_starts(codes, lens, k, torch.device("cpu"), chunk=1 << 10, parts=4)
    assert 1400 in starts and 3200 in starts and 100 not in starts
    c2, l2 = R._cut_kmers_out(codes, lens, starts, k)
    after = kmer_multiset(c2, l2)
    assert set(after) == set(before) and max(after.values()) == 1 and int(l2.min()) >= k
    assert R._duplicate_kmer_starts(c2, l2, k, torch.device("cpu")).size == 0
    # and the generator as a whole: no k-mer twice, at a size where the families alone leave none
    words, endpoints = R.make_recipe_spss("se_k31", 3_000_000, seed=7, device="cpu")
    total = int(endpoints[-1])
    pos = np.arange(total, dtype=np.int64)
    again = torch.from_numpy(((words[pos >> 5] >> ((pos & 31) * 2).astype(np.uint64)) & np.uint64(3)).astype(np.uint8))
    assert R._duplicate_kmer_starts(again, torch.from_numpy(np.diff(endpoints).astype(np.int64)), k, torch.device("cpu")).size == 0


def test_streaming_algorithmic_bytes_follow_the_reference_state_machine(case_se_regular):
    """oracle_streaming_count_bytes (bench.py's roofline for the streaming lines): 1 B per base + 8 B per distinct index word the
    reference's state machine dereferences per k-mer. An extension costs the strings' next k-mer (one or two words), a seed that the
    unchanged-minimizer test cuts short costs nothing, a read shorter than k costs its bases."""
    case = case_se_regular
    s = max(case.sequences, key=len)
    whole = s[500:800]                                   # one search, then extensions only
    rep = case.oracle.streaming_query([whole])
    assert rep["num_searches"] == 1 and rep["num_extensions"] == len(whole) - case.k
    b = case.oracle.streaming_count_bytes([whole])
    assert len(whole) + 8 * rep["num_extensions"] <= b <= len(whole) + 16 * rep["num_extensions"] + 400
    assert case.oracle.streaming_count_bytes(["ACGT"]) == 4 and case.oracle.streaming_count_bytes([]) == 0
    poly = "A" * 200                                     # one minimizer throughout: after the first seed every k-mer is cut short (if absent)
    rep = case.oracle.streaming_query([poly])
    if rep["num_negative_kmers"] == rep["num_kmers"]:
        assert case.oracle.streaming_count_bytes([poly]) < len(poly) + 400
    assert case.oracle.streaming_count_bytes([whole, poly]) == b + case.oracle.streaming_count_bytes([poly])
