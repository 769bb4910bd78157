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
    if c.k != 63:
        pytest.skip("the C4-shaped case")
    dev = torch.device("cuda", 0)
    R, L, parts = 100_000_000, 150, 10
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
    assert whole[1] > 0.2 * whole[0] and whole[3] > 0 and whole[5] > whole[4]  # hits, N's, mostly extensions
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


def test_full_size_human_scale_dictionary_properties():
    """BASELINE.json configs[2] at FULL size: the dictionary bench.py indexes by default (built here, or taken from the
    bench's cache), 10^8 strided ids: lookup(access(id)) == id on both strands, is_member, two launches identical, and
    10^8 random negatives all absent."""
    import torch

    import bench
    from sshash_amd.synthetic import revcomp_device

    bases, recipe, _, _ = bench.WORKLOADS["c3"]
    args = argparse.Namespace(bases=bases, k=31, m=21, recipe=recipe, repeat_scale=1.0, canonical=False, seed=0x5555AAAA,
                              cache_dir=os.environ.get("SSHASH_BENCH_CACHE", "/tmp"), verbose=False)
    d, _ = bench.get_index(args, 0, 1, lambda: None)
    d.to_device(0)
    assert d.num_kmers() > 2_400_000_000
    stats = d.device_stats(0)
    assert stats["sk_slots"] > 0, "the human-scale dictionary must be served by the super-k-mer table"
    dev = torch.device("cuda", 0)
    n = 100_000_000
    stride = d.num_kmers() // n
    ids = torch.arange(n, dtype=torch.int64, device=dev) * stride + 7
    q = torch.empty((n, 1), dtype=torch.int64, device=dev)
    d.access_packed_device(0, ids.data_ptr(), n, q.data_ptr())
    out = torch.empty(n, dtype=torch.int64, device=dev)
    again = torch.empty(n, dtype=torch.int64, device=dev)
    member = torch.empty(n, dtype=torch.uint8, device=dev)
    for qq in (q, revcomp_device(q, 31).contiguous()):
        d.lookup_device(0, qq.data_ptr(), n, out.data_ptr())
        d.lookup_device(0, qq.data_ptr(), n, again.data_ptr())
        d.is_member_device(0, qq.data_ptr(), n, member.data_ptr())
        torch.cuda.synchronize()
        assert int((out != ids).sum().item()) == 0
        assert torch.equal(out, again)
        assert int((member != 1).sum().item()) == 0
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    neg = ((torch.randint(0, 1 << 31, (n,), generator=g, device=dev, dtype=torch.int64) << 31)
           | torch.randint(0, 1 << 31, (n,), generator=g, device=dev, dtype=torch.int64))
    d.lookup_device(0, neg.data_ptr(), n, out.data_ptr())
    torch.cuda.synchronize()
    # a uniformly random 31-mer IS in the dictionary with probability 2 * 2.5e9 / 4^31 ~ 1e-9: 0.1 expected among 10^8;
    # whatever is reported found must really be there
    found = torch.nonzero(out != -1)[:, 0]
    assert found.numel() <= 3
    if found.numel():
        back = torch.empty((found.numel(), 1), dtype=torch.int64, device=dev)
        hit_ids = out[found].contiguous()
        d.access_packed_device(0, hit_ids.data_ptr(), found.numel(), back.data_ptr())
        torch.cuda.synchronize()
        asked = neg[found].unsqueeze(1)
        assert bool(((back == asked) | (revcomp_device(back, 31) == asked)).all().item())
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

