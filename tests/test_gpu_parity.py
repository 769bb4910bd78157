"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the input-order ground truth.

Integer/index work: every comparison is bit-exact.
"""
from __future__ import annotations

import os

import numpy as np
import pytest

import sshash_amd
from conftest import ALL_SMALL_CASES, SE_FASTA, random_dna

pytestmark = pytest.mark.gpu

U64_FIELDS = ["kmer_id", "kmer_id_in_string", "kmer_offset", "string_id", "string_begin", "string_end"]


def _check_full(case, queries, check_rc=True):
    d = case.dict.to_device(0)
    assert d.device_stats()["sk_slots"] > 0  # every k and flavour gets the super-k-mer table
    got = d.lookup(queries, check_reverse_complement=check_rc, full=True)
    want = case.oracle.lookup_packed(queries, check_rc)
    for f in U64_FIELDS:
        assert (getattr(got, f) == want[f]).all(), f"{case.name}: field {f} differs from the oracle"
    assert (got.kmer_orientation.astype(np.int64) == want["kmer_orientation"]).all(), "orientation differs"
    assert (got.minimizer_found == want["minimizer_found"]).all(), "minimizer_found differs"
    # ids-only kernel variant and is_member agree with the full one
    ids = d.lookup(queries, check_reverse_complement=check_rc).kmer_id
    assert (ids == want["kmer_id"]).all()
    mem = d.is_member(queries, check_reverse_complement=check_rc)
    assert (mem == (want["kmer_id"] != sshash_amd.INVALID_U64)).all()
    return got, want


@pytest.mark.parametrize("case_name", ["case_se_regular", "case_se_canonical"] + ALL_SMALL_CASES + ["case_k63_regular"])
def test_lookup_matches_oracle_and_ground_truth(case_name, request):
    case = request.getfixturevalue(case_name)
    n = 60000 if case.gt.num_kmers > 100000 else 4000
    queries = case.queries(n, n, seed=11)
    got, want = _check_full(case, queries)
    g = case.gt.lookup(queries)
    for f in U64_FIELDS:
        assert (getattr(got, f) == g[f]).all(), f"{case.name}: field {f} differs from the ground truth"
    found = g["found"]
    assert (got.kmer_orientation[found] == g["kmer_orientation"][found]).all()
    assert found.sum() >= n  # every positive was found (negatives may collide only by astronomic luck)


@pytest.mark.parametrize("case_name", ["case_se_regular", "case_skew_regular"])
def test_no_reverse_complement_check(case_name, request):
    case = request.getfixturevalue(case_name)
    queries = case.queries(3000, 3000, seed=2)
    got, want = _check_full(case, queries, check_rc=False)
    # half of the positives were reverse-complemented: they must now be misses
    assert (got.kmer_id == sshash_amd.INVALID_U64).sum() >= 3000 + 1400


@pytest.mark.parametrize("case_name", ["case_skew_regular", "case_skew_canonical", "case_k63_canonical", "case_small_k"])
def test_every_kmer_of_the_input_in_file_order(case_name, request):
    """The reference's own contract, test/check_from_file.hpp:38-83: stream the build input, lower-case
    every other sequence, reverse-complement every other k-mer: ids must be 0,1,2,..."""
    case = request.getfixturevalue(case_name)
    d = case.dict.to_device(0)
    k = case.k
    comp = str.maketrans("ACGTacgt", "TGCAtgca")
    kmers, orient = [], []
    count = 0
    for s_id, s in enumerate(case.sequences):
        if s_id % 2 == 0:
            s = s.lower()
        for i in range(len(s) - k + 1):
            x = s[i:i + k]
            if count % 2 == 0:
                x = x.translate(comp)[::-1]
                orient.append(-1)
            else:
                orient.append(1)
            kmers.append(x)
            count += 1
    res = d.lookup(kmers, full=True)
    assert (res.kmer_id == np.arange(count, dtype=np.uint64)).all(), "wrong id assigned"
    assert (res.kmer_orientation == np.array(orient, dtype=np.int8)).all()
    # every output mode is its own kernel instance: all of them over every k-mer of the input
    assert (d.lookup(kmers).kmer_id == res.kmer_id).all(), "ids-only kernel"
    assert d.is_member(kmers).all(), "membership kernel"
    packed = case.gt.kmers(np.arange(count))
    assert d.is_member(packed).all() and (d.lookup(packed).kmer_id == res.kmer_id).all(), "packed entry points"
    partial = d.lookup(packed, full=True)  # (host path asks for every field: the complete kernel)
    assert (partial.kmer_offset == res.kmer_offset).all()
    sizes = res.string_end - res.string_begin - np.uint64(k - 1)
    assert (res.kmer_id_in_string < sizes).all()
    # access round trip: access(lookup(x).kmer_id) is x or its reverse complement (:146-155)
    for i in range(0, count, max(1, count // 200)):
        back = d.access(int(res.kmer_id[i]))
        assert back.upper() in (kmers[i].upper(), kmers[i].upper().translate(comp)[::-1])


def test_ascii_entry_point_ignores_case_and_needs_no_terminator(case_skew_regular):
    case = case_skew_regular
    d = case.dict.to_device(0)
    s = case.sequences[3]
    kmers = [s[i:i + case.k] for i in range(len(s) - case.k + 1)]
    up = d.lookup(kmers).kmer_id
    low = d.lookup([x.lower() for x in kmers]).kmer_id
    assert (up == low).all() and (up != sshash_amd.INVALID_U64).all()
    # n*k bytes back to back, no NUL: reference reads exactly k chars (src/dictionary.cpp:58-63)
    flat = np.frombuffer("".join(kmers).encode(), dtype=np.uint8)
    assert (d.lookup(flat).kmer_id == up).all()
    want = case.oracle.lookup_ascii(flat)
    assert (want["kmer_id"] == up).all()


def test_invalid_characters_are_silently_mapped(case_skew_regular):
    """lookup(char const*) does not validate (only streaming_query does): (c>>1)&3 maps any byte."""
    case = case_skew_regular
    d = case.dict.to_device(0)
    s = case.sequences[0][: case.k]
    weird = "N" + s[1:]  # 'N' = 0x4E -> (0x4E>>1)&3 = 3 = G
    as_g = "G" + s[1:]
    a = d.lookup([weird], full=True)
    b = d.lookup([as_g], full=True)
    assert a.kmer_id[0] == b.kmer_id[0]
    assert (case.oracle.lookup_ascii(np.frombuffer(weird.encode(), dtype=np.uint8))["kmer_id"][0] == a.kmer_id[0])


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 255, 257, 1000])
def test_empty_and_ragged_batches(case_skew_regular, n):
    case = case_skew_regular
    d = case.dict.to_device(0)
    queries = case.queries(n - n // 2, n // 2, seed=n)
    got = d.lookup(queries).kmer_id
    assert got.shape == (n,)
    if n:
        assert (got == case.oracle.lookup_ids(queries)).all()


def test_full_size_roundtrip_every_kmer(case_se_regular, case_se_canonical):
    """test/check.hpp:29-49 at full size: lookup(access(id)).kmer_id == id for EVERY id (4.79 M k-mers),
    via device buffers, plus a checksum of the reverse-complemented batch."""
    import torch

    for case in (case_se_regular, case_se_canonical):
        d = case.dict.to_device(0)
        n = d.num_kmers()
        packed = d.access_packed(np.arange(n, dtype=np.uint64))
        assert (packed == case.gt.kmers(np.arange(n))).all()
        dq = torch.from_numpy(packed.view(np.int64)).cuda()
        out = torch.empty(n, dtype=torch.int64, device="cuda")
        ori = torch.empty(n, dtype=torch.int8, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        d.lookup_device(0, dq.data_ptr(), n, out.data_ptr(), stream=s, kmer_orientation=ori.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(out, torch.arange(n, dtype=torch.int64, device="cuda"))
        assert bool((ori == 1).all())
        from oracle.ground_truth import _revcomp_u64

        rc = torch.from_numpy(_revcomp_u64(packed, case.k).view(np.int64)).cuda()
        d.lookup_device(0, rc.data_ptr(), n, out.data_ptr(), stream=s, kmer_orientation=ori.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(out, torch.arange(n, dtype=torch.int64, device="cuda"))
        assert bool((ori == -1).all())


def test_negative_batch_full_size(case_se_regular):
    """test/check.hpp:78-96 with a real ground truth: 1M uniform random 31-mers, none may be found
    unless the ground truth says so."""
    case = case_se_regular
    d = case.dict.to_device(0)
    rng = np.random.default_rng(123)
    q = rng.integers(0, 1 << 62, 1_000_000, dtype=np.uint64)
    got = d.lookup(q).kmer_id
    g = case.gt.lookup(q)
    assert (got == g["kmer_id"]).all()


def test_index_file_roundtrip_on_gpu(case_skew_canonical, tmp_path):
    case = case_skew_canonical
    d2 = sshash_amd.Dictionary.load(case.index_path).to_device(0)
    q = case.queries(2000, 2000, seed=9)
    assert (d2.lookup(q).kmer_id == case.oracle.lookup_ids(q)).all()
    assert (d2.k(), d2.m(), d2.canonical(), d2.num_kmers()) == (case.k, case.m, True, case.gt.num_kmers)


def test_access_on_device(case_se_regular, case_k63_canonical):
    """access(kmer_id) as a device kernel equals the host access / the ground truth."""
    import torch

    for case in (case_se_regular, case_k63_canonical):
        d = case.dict.to_device(0)
        rng = np.random.default_rng(3)
        ids = rng.integers(0, case.gt.num_kmers, 50000, dtype=np.uint64)
        ids[:3] = [0, case.gt.num_kmers - 1, case.gt.num_kmers]  # last entry is out of range
        d_ids = torch.from_numpy(ids.view(np.int64)).cuda()
        out = torch.empty(ids.size * case.W, dtype=torch.int64, device="cuda")
        d.access_packed_device(0, d_ids.data_ptr(), ids.size, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        got = out.cpu().numpy().view(np.uint64).reshape(-1, case.W)
        assert (got[2] == np.uint64(0xFFFFFFFFFFFFFFFF)).all()
        ok = np.ones(ids.size, dtype=bool)
        ok[2] = False
        assert (got[ok].reshape(-1) == case.gt.kmers(ids[ok])).all()


@pytest.mark.parametrize("case_name", ["case_se_regular", "case_se_canonical", "case_k63_regular", "case_skew_canonical"])
def test_repeated_launches_are_deterministic(case_name, request):
    """Every id / membership bit of every k-mer, several launches in a row: identical and correct each time
    (guards the multi-pass lookup: settled lanes, resumed lanes, deferred lanes, directory overflow)."""
    case = request.getfixturevalue(case_name)
    d = case.dict.to_device(0)
    n = d.num_kmers()
    q = d.access_packed(np.arange(n, dtype=np.uint64))
    want = np.arange(n, dtype=np.uint64)
    for _ in range(4):
        assert (d.lookup(q).kmer_id == want).all()
        assert d.is_member(q).all()


@pytest.mark.parametrize("case_name", ["case_se_regular", "case_skew_canonical", "case_k63_canonical"])
def test_full_result_without_minimizer_found_uses_the_same_values(case_name, request):
    """Asking for every field except `minimizer_found` goes through the multi-pass kernels; the values
    must equal the oracle's all the same."""
    import torch

    case = request.getfixturevalue(case_name)
    d = case.dict.to_device(0)
    n = 20000 if case.gt.num_kmers > 100000 else 3000
    # a random mix, then every k-mer of the dictionary (each super-k-mer, list and deferred key at least once)
    every = case.gt.kmers(np.arange(min(case.gt.num_kmers, 1_000_000)))
    for q in (case.queries(n, n, seed=21), every):
        total = q.size // case.W
        want = case.oracle.lookup_packed(q)
        dq = torch.from_numpy(q.view(np.int64)).cuda()
        bufs = {f: torch.empty(total, dtype=torch.int64, device="cuda") for f in U64_FIELDS}
        ori = torch.empty(total, dtype=torch.int8, device="cuda")
        extra = {f: t.data_ptr() for f, t in bufs.items() if f != "kmer_id"}
        d.lookup_device(0, dq.data_ptr(), total, bufs["kmer_id"].data_ptr(), stream=torch.cuda.current_stream().cuda_stream,
                        kmer_orientation=ori.data_ptr(), **extra)
        torch.cuda.synchronize()
        for f in U64_FIELDS:
            assert (bufs[f].cpu().numpy().view(np.uint64) == want[f]).all(), f
        assert (ori.cpu().numpy().astype(np.int64) == want["kmer_orientation"]).all()


def test_batch_larger_than_one_launch_piece(case_se_regular):
    """More than 2^27 queries in one call: the multi-pass path splits the batch into launch sequences; ids at
    and around the seam must be right (positives planted there, oracle on a sample of the rest)."""
    import torch

    case = case_se_regular
    d = case.dict.to_device(0)
    n = (1 << 27) + 70001
    rng = np.random.default_rng(77)
    q = rng.integers(0, 1 << 62, n, dtype=np.uint64)
    ids = rng.integers(0, case.gt.num_kmers, 200000, dtype=np.uint64)
    where = np.concatenate([np.arange((1 << 27) - 50000, (1 << 27) + 50000), rng.choice((1 << 27) - 50000, 100000, replace=False)])
    q[where] = case.gt.kmers(ids)
    dq = torch.from_numpy(q.view(np.int64)).cuda()
    out = torch.full((n,), -3, dtype=torch.int64, device="cuda")
    d.lookup_device(0, dq.data_ptr(), n, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint64)
    assert (got[where] == ids).all()
    sample = np.concatenate([np.arange(0, n, 997), np.arange((1 << 27) - 3000, (1 << 27) + 3000)])
    assert (got[sample] == case.oracle.lookup_ids(q[sample])).all()
    assert int((got == np.uint64(0xFFFFFFFFFFFFFFFD)).sum()) == 0  # nothing left unwritten


@pytest.mark.parametrize("settings", [{"SSHASH_AMD_DIRECTORY": "0", "SSHASH_AMD_SKTABLE": "0"}, {"SSHASH_AMD_SKTABLE": "0"}, {},
                                      {"SSHASH_AMD_DIRECTORY": "1"}, {"SSHASH_AMD_TEST_HOOKS": "slots_per_key=1.2,slots_per_kmer=1.2"}, {"SSHASH_AMD_TEST_HOOKS": "piece=4096"},
                                      {"SSHASH_AMD_TEST_HOOKS": "piece=4096", "SSHASH_AMD_SKTABLE": "0"}],
                         ids=["mphf_only", "directory_only", "sktable_lean", "sktable_over_directory", "sktable_packed_tight",
                              "many_launch_pieces", "many_launch_pieces_no_table"])
def test_accelerators_disabled(tmp_path, settings):
    """The lookup structures are layered (device_layout.hpp (3)-(5)): with the super-k-mer table and/or
    the minimizer directory switched off (also what an over-wide dictionary gets) the remaining path must
    give the same ids / membership / full results as the oracle. sktable_lean is the default replica: the table over
    bit-packed codewords, no directory; sktable_over_directory forces the round-2 layout; sktable_packed_tight fills both
    regions of the table (the keys', the heavy keys' k-mers') to a load factor of 0.83: long bucket sequences, and items
    that find no slot and are left to the complete path; many_launch_pieces cuts every batch into sequences of 4096 queries:
    the two sets of queues used in turn, the deferred pass of one sequence on the auxiliary stream beside the next one's
    first pass (k <= 31 with the table), everything in order on the caller's stream otherwise."""
    import os
    import subprocess
    import sys
    import textwrap

    from conftest import ROOT

    script = tmp_path / "layers.py"
    script.write_text(textwrap.dedent(
        """
        import os, sys, tempfile
        import numpy as np
        sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
        import conftest as c
        import sshash_amd
        with tempfile.TemporaryDirectory() as tmp:
            for canonical, seed in ((False, 3), (True, 5)):
                case = c.Case("layers%d" % seed, c.skewed_sequences(31, 11, seed=seed), 31, 11, canonical, tmp)
                d = case.dict.to_device(0)
                stats = d.device_stats()
                table = os.environ.get("SSHASH_AMD_SKTABLE") != "0"
                want_directory = os.environ.get("SSHASH_AMD_DIRECTORY") == "1" or (not table and os.environ.get("SSHASH_AMD_DIRECTORY") != "0")
                assert (stats["directory_sectors"] != 0) == want_directory, stats
                assert (stats["sk_slots"] != 0) == table, stats
                if "slots_per_kmer" in os.environ.get("SSHASH_AMD_TEST_HOOKS", ""):
                    assert stats["sk_heavy_kmers"] > 0 and stats["sk_load_factor"] > 0.75 and stats["sk_deferred_keys"] > 0, stats
                q = case.queries(4000, 4000, seed=1)
                want = case.oracle.lookup_ids(q)
                assert (d.lookup(q).kmer_id == want).all()
                assert (d.is_member(q) == (want != np.uint64(0xFFFFFFFFFFFFFFFF))).all()
                full, ora = d.lookup(q, full=True), case.oracle.lookup_packed(q, True)
                for f in ("kmer_id", "kmer_id_in_string", "kmer_offset", "string_id", "string_begin", "string_end", "minimizer_found"):
                    assert (getattr(full, f) == ora[f]).all(), f
                assert (full.kmer_orientation.astype(np.int64) == ora["kmer_orientation"]).all()
                n = case.gt.num_kmers
                allq = case.gt.kmers(np.arange(n))
                assert (d.lookup(allq).kmer_id == np.arange(n, dtype=np.uint64)).all()
        print("LAYERS OK")
        """))
    env = dict(os.environ)
    env.pop("SSHASH_AMD_DIRECTORY", None)
    env.pop("SSHASH_AMD_SKTABLE", None)
    env.pop("SSHASH_AMD_TEST_HOOKS", None)
    env.update(settings)
    p = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "LAYERS OK" in p.stdout, p.stdout + p.stderr


def _expand_neighbours(q, k, W):
    """Independent expansion of kmer_neighbours' 8 queries (src/dictionary.cpp:111-126) with Python integers."""
    out = np.empty(q.size * 8, dtype=np.uint64)
    mask = (1 << (2 * k)) - 1
    codes = [0, 1, 2, 3]  # entry c = the character with code c: A C T G (reference alphabet, include/kmer.hpp:118)
    for i in range(q.size // W):
        x = int(q[i * W]) | (int(q[i * W + 1]) << 64 if W == 2 else 0)
        for c in range(4):
            f = (x >> 2) | (codes[c] << (2 * (k - 1)))
            b = ((x << 2) & mask) | codes[c]
            for which, y in ((c, f), (4 + c, b)):
                for j in range(W):
                    out[(8 * i + which) * W + j] = (y >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
    return out


@pytest.mark.parametrize("case_name", ["case_skew_regular", "case_skew_canonical", "case_k63_canonical", "case_se_regular"])
def test_neighbours_match_eight_oracle_lookups(case_name, request):
    case = request.getfixturevalue(case_name)
    d = case.dict.to_device(0)
    q = case.queries(1500, 300, seed=77)
    expanded = _expand_neighbours(q, case.k, case.W)
    want = case.oracle.lookup_packed(expanded)
    got = d.neighbours(q, full=True)
    for f in U64_FIELDS:
        assert (getattr(got, f) == want[f]).all(), f
    assert (got.kmer_orientation.astype(np.int64) == want["kmer_orientation"]).all()
    assert (got.minimizer_found.astype(np.int64) == want["minimizer_found"]).all()
    assert (d.neighbours(q, check_reverse_complement=False).kmer_id == case.oracle.lookup_packed(expanded, False)["kmer_id"]).all()
    # positives drawn from inside a string have at least one forward or backward neighbour in the dictionary
    ids = got.kmer_id.reshape(-1, 8)
    assert (ids != np.uint64(0xFFFFFFFFFFFFFFFF)).any()
    # device-pointer form
    import torch

    n = q.size // case.W
    dq = torch.from_numpy(q.view(np.int64)).cuda()
    out = torch.empty(8 * n, dtype=torch.int64, device="cuda")
    d.neighbours_device(0, dq.data_ptr(), n, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert (out.cpu().numpy().view(np.uint64) == want["kmer_id"]).all()


def test_weights_on_the_device_equal_the_abundances_of_the_input_file(tmp_path):
    """dictionary::weight (reference test/check_from_file.hpp:229-275): the id-th weight of the file, in file
    order, for every k-mer; device kernel against the oracle restatement and the file itself."""
    import gzip

    import torch

    from conftest import WEIGHTED_FASTA
    from oracle import oracle as O

    d = sshash_amd.Dictionary.build(WEIGHTED_FASTA, k=31, m=15, weighted=True, num_threads=8).to_device(0)
    path = str(tmp_path / "w.sshash")
    d.save(path)
    want = []
    with gzip.open(WEIGHTED_FASTA, "rt") as f:
        for header in f:
            next(f)
            want.extend(int(x) for x in header.split("ab:Z:")[1].split())
    want = np.array(want, dtype=np.uint64)
    n = d.num_kmers()
    assert n == want.size
    ids = torch.arange(n + 3, dtype=torch.int64, device="cuda")  # three ids past the end
    out = torch.empty(n + 3, dtype=torch.int64, device="cuda")
    d.weight_device(0, ids.data_ptr(), n + 3, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint64)
    assert (got[:n] == want).all()
    assert (got[n:] == sshash_amd.INVALID_U64).all()
    assert (O.OracleIndex(path).weights(np.arange(n, dtype=np.uint64)) == want).all()
    # weights ride along with lookups: lookup -> id -> weight for reverse-complemented queries
    rng = np.random.default_rng(3)
    pick = rng.integers(0, n, 5000)
    packed = d.access_packed(pick)
    assert (d.weight(d.lookup(packed).kmer_id) == want[pick]).all()
    # a dictionary without weights refuses
    plain = sshash_amd.Dictionary.build(WEIGHTED_FASTA, k=31, m=15, num_threads=8).to_device(0)
    with pytest.raises(sshash_amd.SSHashError):
        plain.weight_device(0, ids.data_ptr(), 4, out.data_ptr())


@pytest.mark.parametrize("case_name", ["case_skew_regular", "case_k63_canonical"])
def test_string_neighbours(case_name, request):
    """string_neighbours(s) = forward part of kmer_neighbours(last k-mer of s) + backward part of
    kmer_neighbours(first k-mer of s) (src/dictionary.cpp:189-201)."""
    case = request.getfixturevalue(case_name)
    d = case.dict.to_device(0)
    k, W = case.k, case.W
    sids = np.arange(0, len(case.sequences), 3)
    got = d.string_neighbours(sids).reshape(-1, 8)
    first_ids, last_ids = [], []
    kid = 0
    starts = []
    for s in case.sequences:
        starts.append(kid)
        kid += len(s) - k + 1
    for s in sids:
        first_ids.append(starts[s])
        last_ids.append(starts[s] + len(case.sequences[s]) - k)
    want_f = case.oracle.lookup_packed(_expand_neighbours(case.gt.kmers(np.array(last_ids)), k, W))["kmer_id"].reshape(-1, 8)
    want_b = case.oracle.lookup_packed(_expand_neighbours(case.gt.kmers(np.array(first_ids)), k, W))["kmer_id"].reshape(-1, 8)
    assert (got[:, :4] == want_f[:, :4]).all() and (got[:, 4:] == want_b[:, 4:]).all()
    with pytest.raises(sshash_amd.SSHashError):
        d.string_neighbours([len(case.sequences)])


def test_concurrent_host_callers_on_one_handle(case_se_regular):
    """The query entry points are re-entrant on a handle (const methods in the reference): four host threads
    looking up different batches at once (ctypes drops the GIL) get the answers of sequential calls."""
    import threading

    case = case_se_regular
    d = case.dict.to_device(0)
    batches = [case.queries(150000, 150000, seed=100 + t) for t in range(4)]
    want = [case.oracle.lookup_ids(q) for q in batches]
    got = [None] * 4
    errors = []

    def work(t):
        try:
            for _ in range(3):
                got[t] = d.lookup(batches[t]).kmer_id
                assert (d.is_member(batches[t]) == (got[t] != sshash_amd.INVALID_U64)).all()
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for t in range(4):
        assert (got[t] == want[t]).all()


def test_concurrent_device_callers_on_the_null_stream(case_se_regular):
    """Two host threads may share a stream (the null stream): their launch sequences share that stream's
    deferred-query scratch and must be enqueued one after the other, never interleaved."""
    import threading

    import torch

    case = case_se_regular
    d = case.dict.to_device(0)
    batches = [case.queries(200000, 200000, seed=300 + t) for t in range(4)]
    want = [case.oracle.lookup_ids(q) for q in batches]
    dq = [torch.from_numpy(q.view(np.int64)).cuda() for q in batches]
    out = [torch.empty(q.size, dtype=torch.int64, device="cuda") for q in batches]
    torch.cuda.synchronize()

    def work(t):
        for _ in range(20):
            d.lookup_device(0, dq[t].data_ptr(), batches[t].size, out[t].data_ptr(), stream=0)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    torch.cuda.synchronize()
    for t in range(4):
        assert (out[t].cpu().numpy().view(np.uint64) == want[t]).all()


@pytest.mark.parametrize("case_name", ["case_skew_regular", "case_skew_canonical", "case_k63_canonical"])
def test_navigational_queries_follow_the_input_strings(case_name, request):
    """The reference's own check (test/check_from_file.hpp:174-221): for every k-mer of the input, forward[code of
    the next base] and backward[code of the previous base] must be in the dictionary -- here also with the id the
    neighbour has in file order (i + 1 and i - 1)."""
    case = request.getfixturevalue(case_name)
    d = case.dict.to_device(0)
    k = case.k
    n = case.gt.num_kmers
    ids = d.neighbours(case.gt.kmers(np.arange(n))).kmer_id.reshape(n, 8)
    code = {"A": 0, "C": 1, "T": 2, "G": 3}
    i = 0
    for s in case.sequences:
        for j in range(len(s) - k + 1):
            if j + k < len(s):
                assert ids[i, code[s[j + k]]] == i + 1
            if j > 0:
                assert ids[i, 4 + code[s[j - 1]]] == i - 1
            i += 1
    assert i == n


@pytest.mark.parametrize("case_name", ["case_se_regular", "case_se_canonical", "case_k63_regular"])
def test_a_full_resume_queue_sends_its_queries_down_the_complete_path(case_name, request, monkeypatch):
    """The second pass's queue holds half a launch piece; a query that finds it full is handed to the complete path
    (directory / MPHF / skew index) instead. With the queue cut to 1/64 nearly every resumed query takes that route:
    ids and membership of every k-mer, both strands, plus a random mix against the oracle, must not change."""
    case = request.getfixturevalue(case_name)
    d = case.dict.to_device(0)
    monkeypatch.setenv("SSHASH_AMD_TEST_HOOKS", "resume_divisor=64")
    n = d.num_kmers()
    q = d.access_packed(np.arange(n, dtype=np.uint64))
    want = np.arange(n, dtype=np.uint64)
    assert (d.lookup(q).kmer_id == want).all()
    assert d.is_member(q).all()
    rc = case.gt._revcomp(q).reshape(-1)
    assert (d.lookup(rc).kmer_id == want).all()
    mix = case.queries(20000, 20000, seed=41)
    assert (d.lookup(mix).kmer_id == case.oracle.lookup_ids(mix)).all()


def test_entry_points_that_launch_once_refuse_what_one_launch_cannot_carry(case_se_regular):
    """A launch of 2^32 threads or more is silently not carried out (HISTORY.md): access / weight / neighbours /
    route say so instead of returning untouched output. (Nothing is dereferenced before the check.)"""
    import torch

    d = case_se_regular.dict.to_device(0)
    buf = torch.zeros(16, dtype=torch.int64, device="cuda:0")
    with pytest.raises(sshash_amd.SSHashError, match="2\\^32"):
        d.access_packed_device(0, buf.data_ptr(), 1 << 32, buf.data_ptr())
    d.access_packed_device(0, buf.data_ptr(), 16, buf.data_ptr())  # (ids 0: fine)
    torch.cuda.synchronize()


def test_page_locked_caller_buffers_are_used_in_place(case_se_regular):
    """Host entry points: page-locked input and output are copied from and to directly (no staging through the library's
    lanes); same ids as with pageable buffers, also when only one side is page-locked (then both are staged)."""
    import torch

    case = case_se_regular
    d = case.dict.to_device(0)
    q = case.queries(150_000, 150_000, seed=41)
    want = d.lookup(q).kmer_id
    q_pin = torch.from_numpy(q.view(np.int64)).pin_memory()
    out_pin = torch.empty(want.size, dtype=torch.int64).pin_memory()
    out_pin.fill_(7)
    d.lookup(q_pin.numpy().view(np.uint64), out=out_pin.numpy().view(np.uint64))
    assert (out_pin.numpy().view(np.uint64) == want).all()
    mixed = np.full(want.size, 7, dtype=np.uint64)
    d.lookup(q_pin.numpy().view(np.uint64), out=mixed)
    assert (mixed == want).all()
    assert (want == case.oracle.lookup_packed(q, True)["kmer_id"]).all()
    # round 6: device-mapped page-locked arrays are read and written by the kernels where they lie (the call above); the copy pipeline
    # over the same arrays (what arrays that are page-locked but not mapped take), all eight fields and is_member both ways
    full_mapped = d.lookup(q_pin.numpy().view(np.uint64), full=True, out=out_pin.numpy().view(np.uint64))
    os.environ["SSHASH_AMD_TEST_HOOKS"] = "host_staged_copies=1,host_chunk=65536"
    try:
        out_pin.fill_(7)
        d.lookup(q_pin.numpy().view(np.uint64), out=out_pin.numpy().view(np.uint64))
        assert (out_pin.numpy().view(np.uint64) == want).all()
    finally:
        del os.environ["SSHASH_AMD_TEST_HOOKS"]
    full_pageable = d.lookup(q, full=True)
    for f in ("kmer_id", "kmer_id_in_string", "kmer_offset", "string_id", "string_begin", "string_end", "kmer_orientation", "minimizer_found"):
        assert (getattr(full_mapped, f) == getattr(full_pageable, f)).all(), f


def test_many_caller_streams(case_se_regular):
    """A stream per request: the per-stream queue scratch is capped (the least recently used stream's block is freed when a
    17th stream shows up), and every stream -- new, evicted and back again -- gets right answers."""
    import torch

    case = case_se_regular
    d = case.dict.to_device(0)
    q = case.queries(20_000, 20_000, seed=43)
    want = torch.from_numpy(case.oracle.lookup_packed(q, True)["kmer_id"].view(np.int64)).cuda()
    dq = torch.from_numpy(q.view(np.int64)).cuda()
    streams = [torch.cuda.Stream() for _ in range(40)]
    outs = [torch.empty(want.numel(), dtype=torch.int64, device="cuda:0") for _ in streams]
    torch.cuda.synchronize()
    for turn in range(2):
        for s, o in zip(streams, outs):
            o.fill_(5)
            torch.cuda.synchronize()
            d.lookup_device(0, dq.data_ptr(), want.numel(), o.data_ptr(), stream=s.cuda_stream)
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, want)

