"""GPU parity over the (k, m, canonical) plane: one small dictionary per point, every k-mer of it and as many random
negatives through every kernel instance (ids, membership, full result), against the CPU oracle and the ground truth read
off the input. The table key, the bucket width (k <= 31: 64-byte buckets, k > 31: two lines) and the walker's window
arithmetic all depend on k and m; the fixtures of conftest.py cover eight points, this covers the edges in between
(m = k, m = 31, m small, k = 33 -- the first two-word k --, k = 63)."""
from __future__ import annotations

import numpy as np
import pytest

import sshash_amd
from conftest import Case, random_dna

pytestmark = pytest.mark.gpu

POINTS = [(21, 11), (23, 23), (27, 9), (29, 17), (31, 7), (31, 27), (31, 31), (33, 13), (35, 31), (41, 21), (47, 15),
          (55, 19), (61, 29), (63, 31)]


@pytest.mark.parametrize("canonical", [False, True], ids=["regular", "canonical"])
@pytest.mark.parametrize("k,m", POINTS, ids=[f"k{k}m{m}" for k, m in POINTS])
def test_point(k, m, canonical, tmp_path):
    rng = np.random.default_rng(1000 * k + m + int(canonical))
    # strings of ragged lengths, the shortest exactly one k-mer long
    lengths = [k, k + 1, 2 * k - 1, 2 * k] + [int(x) for x in rng.integers(k, 40 * k, 60)]
    sequences = [random_dna(rng, n) for n in lengths]
    case = Case(f"sweep_k{k}_m{m}_{int(canonical)}", sequences, k, m, canonical, str(tmp_path))
    d = case.dict.to_device(0)
    assert d.device_stats()["sk_slots"] > 0  # the super-k-mer table serves every point
    n = case.gt.num_kmers
    every = case.gt.kmers(np.arange(n)).reshape(n, case.W)
    rc = case.gt._revcomp(every.reshape(-1)).reshape(n, case.W)
    neg = rng.integers(0, 1 << 62, (n, case.W), dtype=np.uint64)
    neg[:, -1] &= np.uint64((1 << (2 * k - 64 * (case.W - 1))) - 1)
    queries = np.ascontiguousarray(np.concatenate([every, rc, neg])).reshape(-1)
    want = case.oracle.lookup_packed(queries, True)
    got = d.lookup(queries, full=True)
    for f in ("kmer_id", "kmer_id_in_string", "kmer_offset", "string_id", "string_begin", "string_end"):
        assert (getattr(got, f) == want[f]).all(), f
    assert (got.kmer_orientation.astype(np.int64) == want["kmer_orientation"]).all()
    assert (got.minimizer_found == want["minimizer_found"]).all()
    ids = d.lookup(queries).kmer_id
    assert (ids == want["kmer_id"]).all()
    assert (ids[:n] == np.arange(n, dtype=np.uint64)).all() and (ids[n:2 * n] == np.arange(n, dtype=np.uint64)).all()
    assert (d.is_member(queries) == (want["kmer_id"] != sshash_amd.INVALID_U64)).all()
    # forward only: the reverse complements are misses unless the k-mer is in the input on that strand too
    fwd = d.lookup(queries, check_reverse_complement=False).kmer_id
    assert (fwd == case.oracle.lookup_packed(queries, False)["kmer_id"]).all()
    # the streaming query over reads cut out of the strings (substitutions, N's, both strands, ends of strings) and random
    # ones: the six counters and every per-k-mer result against the oracle's restated state machine
    from test_gpu_streaming import _as_dict, _synthetic_reads

    reads = _synthetic_reads(case, 400, seed=k + m, read_len=3 * k) + [sequences[-1], sequences[0] + "ACGT" * 5, sequences[2][::-1]]
    want_report = case.oracle.streaming_query(reads)
    assert _as_dict(d.streaming_query(reads)) == want_report
    per_read, report = d.streaming_lookup(reads, full=True)
    assert _as_dict(report) == want_report
    for read, got in zip(reads, per_read):
        want = case.oracle.streaming_read(read)
        assert got.kmer_id.size == want.size == max(0, len(read) - k + 1)
        found = want["kmer_id"] != np.uint64(0xFFFFFFFFFFFFFFFF)
        for f in ("kmer_id", "kmer_id_in_string", "string_id", "string_begin", "string_end"):
            assert (getattr(got, f) == want[f]).all(), f
        assert (got.kmer_orientation[found] == want["kmer_orientation"][found]).all()
    d.close()
