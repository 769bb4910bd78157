"""GPU: the BASELINE.json workloads themselves: the synthetic stand-ins bench.py indexes
(sshash_amd/synthetic.py) at reduced size for k=31 m=21 regular / canonical and k=63 m=25 -- checked against the CPU
oracle and against a ground truth read straight off the packed strings --, and the human-scale C3 dictionary at FULL
size through size-independent properties (reference test/check.hpp:29-49: lookup(access(i)) == i; :78-96: random
negatives)."""
from __future__ import annotations

import argparse
import os

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

INVALID = np.uint64(0xFFFFFFFFFFFFFFFF)


def kmers_at(words: np.ndarray, offsets: np.ndarray, k: int) -> np.ndarray:
    """k-mers starting at the given base offsets of a 2-bit packed array -> (n, W) uint64."""
    W = 1 if k <= 31 else 2
    out = np.zeros((offsets.size, W), dtype=np.uint64)
    w = np.concatenate([words, np.zeros(4, dtype=np.uint64)])
    idx = (offsets >> np.uint64(5)).astype(np.int64)
    sh = ((offsets & np.uint64(31)) * np.uint64(2)).astype(np.uint64)
    inv = (np.uint64(64) - sh) & np.uint64(63)
    def window(j):
        lo, hi = w[idx + j], w[idx + j + 1]
        return np.where(sh == 0, lo, (lo >> sh) | (hi << inv))
    out[:, 0] = window(0)
    if W == 2:
        out[:, 1] = window(1)
        out[:, 1] &= np.uint64((1 << (2 * k - 64)) - 1)
    else:
        out[:, 0] &= np.uint64((1 << (2 * k)) - 1)
    return out


class SyntheticCase:
    def __init__(self, tmpdir, name, bases, k, m, canonical, mean_len):
        import sshash_amd
        from oracle import oracle as O
        from sshash_amd.repeats import make_recipe_spss

        self.k, self.m, self.W = k, m, 1 if k <= 31 else 2
        # the bench's own stand-ins (repeat families fitted to the published bucket statistics), at reduced size
        recipe = "human_k63" if k > 31 else ("se_k31" if mean_len < 100 else "human_k31")
        self.words, self.endpoints = make_recipe_spss(recipe, bases, seed=4242)
        self.dict = sshash_amd.Dictionary.build_from_packed(self.words, self.endpoints, k=k, m=m, canonical=canonical, num_threads=0)
        self.path = os.path.join(tmpdir, name + ".sshash")
        self.dict.save(self.path)
        self.oracle = O.OracleIndex(self.path)
        self.dict.to_device(0)
        # ground truth off the strings: k-mer at base offset o of string s has id o - s*(k-1) (reference
        # include/spectrum_preserving_string_set.hpp:226-228)
        lens = np.diff(self.endpoints).astype(np.int64)
        self.kmers_per_string = lens - (k - 1)
        self.first_id = np.concatenate([[0], np.cumsum(self.kmers_per_string)[:-1]]).astype(np.uint64)

    def sample_truth(self, n, seed):
        rng = np.random.default_rng(seed)
        ids = rng.integers(0, self.dict.num_kmers(), n).astype(np.uint64)
        s = np.searchsorted(self.first_id, ids, side="right") - 1
        offsets = ids + s.astype(np.uint64) * np.uint64(self.k - 1)
        return ids, kmers_at(self.words, offsets, self.k)


@pytest.fixture(scope="module", params=[("c2_like_regular", 20_000_000, 31, 21, False, 85.0),
                                        ("c3_like_canonical", 20_000_000, 31, 21, True, 274.0),
                                        ("c4_like_k63", 24_000_000, 63, 25, False, 274.0)], ids=lambda p: p[0])
def synthetic_case(request, tmp_path_factory):
    return SyntheticCase(str(tmp_path_factory.mktemp("baseline")), *request.param)


def test_every_kmer_round_trips_on_the_device(synthetic_case):
    """lookup(access(i)) == i for EVERY id, forward and reverse-complemented, with is_member agreeing."""
    import torch
    from sshash_amd.synthetic import revcomp_device

    c, d = synthetic_case, synthetic_case.dict
    n = d.num_kmers()
    dev = torch.device("cuda", 0)
    ids = torch.arange(n, dtype=torch.int64, device=dev)
    q = torch.empty((n, c.W), dtype=torch.int64, device=dev)
    d.access_packed_device(0, ids.data_ptr(), n, q.data_ptr())
    out = torch.empty(n, dtype=torch.int64, device=dev)
    member = torch.empty(n, dtype=torch.uint8, device=dev)
    for qq in (q, revcomp_device(q, c.k).contiguous()):
        d.lookup_device(0, qq.data_ptr(), n, out.data_ptr())
        d.is_member_device(0, qq.data_ptr(), n, member.data_ptr())
        torch.cuda.synchronize()
        assert int((out != ids).sum().item()) == 0
        assert int((member != 1).sum().item()) == 0


def test_ground_truth_read_off_the_packed_strings(synthetic_case):
    c, d = synthetic_case, synthetic_case.dict
    ids, kmers = c.sample_truth(300_000, seed=9)
    got = d.lookup(kmers.reshape(-1)).kmer_id
    assert (got == ids).all()


def test_mixed_batches_match_the_oracle(synthetic_case):
    """The bench's own query mixes (random and mutated negatives), ids against the CPU oracle."""
    import torch
    from sshash_amd.synthetic import draw_queries_device

    c, d = synthetic_case, synthetic_case.dict
    n = 1_000_000
    for negatives in ("random", "mutated"):
        dq = draw_queries_device(d, 0, n, 0.5, seed=77, negatives=negatives)
        out = torch.empty(n, dtype=torch.int64, device=dq.device)
        d.lookup_device(0, dq.data_ptr(), n, out.data_ptr())
        torch.cuda.synchronize()
        got = out.cpu().numpy().view(np.uint64)
        want = c.oracle.lookup_ids(dq.cpu().numpy().view(np.uint64), num_threads=8)
        assert (got == want).all(), negatives
        assert 0.45 < float((got != INVALID).mean()) < 0.56


def test_streaming_query_at_the_read_count_of_config_c4(synthetic_case):
    """BASELINE.json configs[3]: k = 63, streaming_query over 10^8 reads of 150 bases (1.5 x 10^10 bases in one call -- past
    every 2^32 boundary a launch or an index could trip over). The counters of the whole equal the sum over ten parts, the
    position-parallel pipeline (counters-only mode) agrees on one part, and num_kmers is reads x (150 - k + 1). Reads are cut
    out of the dictionary's own strings (concatenated: some run across two strings) with substitutions and N's."""
    import torch

    c, d = synthetic_case, synthetic_case.dict
    dev = torch.device("cuda", 0)
    # (round 6: the k = 31 cases too, at a fifth of the reads -- the regular one is the stand-in with the most k-mers under heavy keys, which
    # is where the run kernel's memory of a heavy key acts: counters of the whole = sum over parts = the position-parallel pipeline's)
    R, L, parts = (100_000_000 if c.k == 63 else 20_000_000), 150, 10
    total = int(c.endpoints[-1])
    words = torch.from_numpy(c.words.view(np.int64)).to(dev)
    pos = torch.arange(total, dtype=torch.int64, device=dev)
    codes = (words[pos >> 5] >> ((pos & 31) * 2)) & 3
    text = torch.tensor(list(b"ACTG"), dtype=torch.uint8, device=dev)[codes]  # the strings as characters
    del pos, codes
    bases = torch.empty(R * L, dtype=torch.uint8, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    per = R // parts
    for a in range(0, R, per):
        start = torch.randint(0, total - L, (per,), generator=g, device=dev, dtype=torch.int64)
        chunk = text[(start[:, None] + torch.arange(L, device=dev)[None, :]).reshape(-1)]
        noise = torch.rand(per * L, generator=g, device=dev)
        chunk[noise < 0.004] = ord("N")
        sub = (noise > 0.99).nonzero()[:, 0]
        chunk[sub] = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[torch.randint(0, 4, (sub.numel(),), generator=g, device=dev)]
        bases[a * L:(a + per) * L] = chunk
        del start, chunk, noise, sub
    offsets = torch.arange(R + 1, dtype=torch.int64, device=dev) * L
    whole = torch.zeros(6, dtype=torch.int64, device=dev)
    d.streaming_query_device(0, bases.data_ptr(), offsets.data_ptr(), R, whole.data_ptr())
    torch.cuda.synchronize()
    whole = whole.cpu().numpy()
    assert whole[0] == R * (L - c.k + 1)
    assert whole[0] == whole[1] + whole[2] + whole[3] and whole[1] == whole[4] + whole[5]
    assert whole[1] > 0.15 * whole[0] and whole[3] > 0 and whole[5] > whole[4]  # hits, N's, mostly extensions
    summed = np.zeros(6, dtype=np.int64)
    rel = torch.arange(per + 1, dtype=torch.int64, device=dev) * L
    for a in range(0, R, per):
        part = torch.zeros(6, dtype=torch.int64, device=dev)
        d.streaming_query_device(0, bases.data_ptr() + a * L, rel.data_ptr(), per, part.data_ptr())
        torch.cuda.synchronize()
        summed += part.cpu().numpy()
        if a == 0:  # the same part through encode -> masked lookup -> classify
            again = torch.zeros(6, dtype=torch.int64, device=dev)
            d.streaming_lookup_device(0, bases.data_ptr(), rel.data_ptr(), per, per * L, 0, d_report=again.data_ptr())
            torch.cuda.synchronize()
            assert (again.cpu().numpy() == part.cpu().numpy()).all()
    assert (summed == whole).all()


def full_size_dictionary(workload, canonical=False):
    """the dictionary bench.py indexes for `workload` (`--canonical`: its canonical flavour) -- built here, or taken from the bench's
    cache -- and the index file it came from"""
    import bench
    from sshash_amd.repeats import load_recipe

    bases, recipe, _, _ = bench.WORKLOADS[workload]
    r = load_recipe(recipe)
    args = argparse.Namespace(bases=bases, k=int(r["k"]), m=int(r["m"]), recipe=recipe, repeat_scale=1.0, canonical=canonical, seed=0x5555AAAA,
                              cache_dir=os.environ.get("SSHASH_BENCH_CACHE", "/tmp"), verbose=False)
    d, path = bench.get_index(args, 0, 1, lambda: None)
    d.to_device(0)
    return d, path, args


@pytest.mark.parametrize("workload,canonical,least_kmers", [("c3", False, 2_400_000_000), ("c2", False, 850_000_000), ("c4", False, 2_600_000_000),
                                                            ("c3", True, 2_400_000_000), ("c4", True, 2_600_000_000)],
                         ids=["c3", "c2", "c4", "c3_canonical", "c4_canonical"])
def test_full_size_dictionary_properties(workload, canonical, least_kmers):
    """BASELINE.json configs[2], [1] and [3] at FULL size -- the very dictionaries bench.py measures (C3: human scale k = 31, C2: S. enterica
    pangenome scale, C4: human scale k = 63, two-word k-mers; C3 and C4 also as CANONICAL dictionaries, src/dictionary.cpp:24-56, the flavour
    the reference publishes beside the regular one: benchmarks/results-21-01-26/k31/canon-bench.json): 10^8 strided ids (5 x 10^7 at k = 63):
    lookup(access(id)) == id on both strands with the orientation the strand implies -- +1 for the k-mer as the strings spell it, -1 for its
    reverse complement (test/check.hpp:29-49, test/check_from_file.hpp:100-116) --, is_member, two launches identical, 10^8 random negatives
    all absent (:78-96) -- and the CPU oracle over 10^6 queries of the bench's own 50/50 mix, read from the index file on disk."""
    import torch

    from oracle import oracle as O
    from sshash_amd.synthetic import draw_queries_device, revcomp_device

    d, path, args = full_size_dictionary(workload, canonical)
    k, W = args.k, 1 if args.k <= 31 else 2
    assert d.num_kmers() > least_kmers and d.k() == k and bool(d.canonical()) == canonical
    stats = d.device_stats(0)
    assert stats["sk_slots"] > 0, "the full-size dictionaries must be served by the super-k-mer table"
    dev = torch.device("cuda", 0)
    n = 100_000_000 if W == 1 else 50_000_000
    stride = d.num_kmers() // n
    ids = torch.arange(n, dtype=torch.int64, device=dev) * stride + 7
    q = torch.empty((n, W), dtype=torch.int64, device=dev)
    d.access_packed_device(0, ids.data_ptr(), n, q.data_ptr())
    out = torch.empty(n, dtype=torch.int64, device=dev)
    again = torch.empty(n, dtype=torch.int64, device=dev)
    member = torch.empty(n, dtype=torch.uint8, device=dev)
    orientation = torch.empty(n, dtype=torch.int8, device=dev)
    for strand, qq in ((1, q), (-1, revcomp_device(q, k).contiguous())):
        d.lookup_device(0, qq.data_ptr(), n, out.data_ptr())
        d.lookup_device(0, qq.data_ptr(), n, again.data_ptr(), kmer_orientation=orientation.data_ptr())
        d.is_member_device(0, qq.data_ptr(), n, member.data_ptr())
        torch.cuda.synchronize()
        assert int((out != ids).sum().item()) == 0
        assert torch.equal(out, again)
        assert int((orientation != strand).sum().item()) == 0
        assert int((member != 1).sum().item()) == 0
    del orientation
    del again, member
    g = torch.Generator(device=dev)
    g.manual_seed(11)

    def random_words(count, bits):  # uniform `bits`-bit values as int64 (bits <= 62)
        hi = torch.randint(0, 1 << (bits - 31), (count,), generator=g, device=dev, dtype=torch.int64)
        return (hi << 31) | torch.randint(0, 1 << 31, (count,), generator=g, device=dev, dtype=torch.int64)

    if W == 1:
        neg = random_words(n, 62).unsqueeze(1)
    else:  # k = 63: 126 bits in two words (64 + 62)
        low = random_words(n, 62) ^ (random_words(n, 33) << 31)  # all 64 bits of word 0 vary
        neg = torch.stack([low, random_words(n, 2 * k - 64)], dim=1).contiguous()
    d.lookup_device(0, neg.data_ptr(), n, out.data_ptr())
    torch.cuda.synchronize()
    # a uniformly random 31-mer IS in the dictionary with probability 2 * 2.5e9 / 4^31 ~ 1e-9: 0.1 expected among 10^8 (none at k = 63);
    # whatever is reported found must really be there
    found = torch.nonzero(out != -1)[:, 0]
    assert found.numel() <= 3
    if found.numel():
        back = torch.empty((found.numel(), W), dtype=torch.int64, device=dev)
        hit_ids = out[found].contiguous()
        d.access_packed_device(0, hit_ids.data_ptr(), found.numel(), back.data_ptr())
        torch.cuda.synchronize()
        asked = neg[found]
        assert bool(((back == asked).all(dim=1) | (revcomp_device(back, k) == asked).all(dim=1)).all().item())
    del neg, q, ids
    # the oracle (CPU restatement, reading the index file the dictionary was loaded from) over the bench's own mix
    m = 1_000_000
    dq = draw_queries_device(d, 0, m, 0.5, seed=args.seed + 99)
    d.lookup_device(0, dq.data_ptr(), m, out.data_ptr())
    torch.cuda.synchronize()
    got = out[:m].cpu().numpy().view(np.uint64)
    want = O.OracleIndex(path).lookup_ids(dq.cpu().numpy().view(np.uint64), num_threads=max(1, len(os.sched_getaffinity(0))))
    assert (got == want).all()
    assert 0.49 < float((got != INVALID).mean()) < 0.51
    d.close()


def test_streaming_query_against_the_full_size_k63_dictionary():
    """BASELINE.json configs[3] on the dictionary it names: the streaming query's six counters on 60 000 reads of the bench's own set (half
    from the 2.96 G-base k = 63 dictionary with 1 % substitutions, half random, N at 1e-3) equal the CPU oracle's restated state machine
    (include/streaming_query.hpp:48-197) -- through the run-based kernel and through the position-parallel pipeline."""
    import torch

    from oracle import oracle as O
    from sshash_amd.synthetic import make_reads_device

    d, path, args = full_size_dictionary("c4")
    dev = torch.device("cuda", 0)
    n, L = 60_000, 150
    reads = make_reads_device(d, 0, n, L, positive_fraction=0.5, seed=args.seed + 7)
    offsets = torch.arange(n + 1, dtype=torch.int64, device=dev) * L
    want = O.OracleIndex(path).streaming_query([bytes(r) for r in reads.cpu().numpy()])
    names = ("num_kmers", "num_positive_kmers", "num_negative_kmers", "num_invalid_kmers", "num_searches", "num_extensions")
    assert want["num_kmers"] == n * (L - 63 + 1) and want["num_extensions"] > 10 * want["num_searches"] > 0 and want["num_invalid_kmers"] > 0
    for how in ("runs", "positions"):
        report = torch.zeros(6, dtype=torch.int64, device=dev)
        if how == "positions":
            d.streaming_lookup_device(0, reads.data_ptr(), offsets.data_ptr(), n, n * L, 0, d_report=report.data_ptr())
        else:
            d.streaming_query_device(0, reads.data_ptr(), offsets.data_ptr(), n, report.data_ptr())
        torch.cuda.synchronize()
        got = dict(zip(names, (int(v) for v in report.cpu().tolist())))
        assert got == {f: int(v) for f, v in want.items()}, how
    d.close()


@pytest.mark.skipif(os.environ.get("SSHASH_TEST_HUGE") != "1", reason="several minutes and ~100 GB of HBM: set SSHASH_TEST_HUGE=1")
def test_a_dictionary_with_more_than_2_32_kmer_starts_still_gets_its_table():
    """4.6 x 10^9 bases: the table build's one-lane-per-k-mer-start scans no longer fit one launch (2^32 threads: such a
    launch is silently not carried out) and run in pieces; lookup(access(id)) == id over strided ids, both strands."""
    import torch

    import bench
    from sshash_amd.synthetic import revcomp_device

    args = argparse.Namespace(bases=4_600_000_000, k=31, m=21, canonical=False, seed=0x77AA, recipe="human_k31", repeat_scale=1.0,
                              cache_dir=os.environ.get("SSHASH_BENCH_CACHE", "/tmp"), verbose=False)
    d, _ = bench.get_index(args, 0, 1, lambda: None)
    d.to_device(0)
    stats = d.device_stats(0)
    assert d.num_kmers() > 4_000_000_000 and stats["sk_slots"] > 0 and stats["sk_absent_reason"] is None, stats
    dev = torch.device("cuda", 0)
    n = 50_000_000
    ids = torch.arange(n, dtype=torch.int64, device=dev) * (d.num_kmers() // n) + 3
    q = torch.empty((n, 1), dtype=torch.int64, device=dev)
    d.access_packed_device(0, ids.data_ptr(), n, q.data_ptr())
    out = torch.empty(n, dtype=torch.int64, device=dev)
    for qq in (q, revcomp_device(q, 31).contiguous()):
        d.lookup_device(0, qq.data_ptr(), n, out.data_ptr())
        torch.cuda.synchronize()
        assert int((out != ids).sum().item()) == 0
    d.close()

