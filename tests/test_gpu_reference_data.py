"""GPU: the rest of the reference's own data as parity inputs -- copied verbatim under tests/golden/ by
tests/golden/make_golden.py: the complete k = 63 and k = 47 stitched unitigs (two-word k-mers; at k = 47 the second word is
half used), E. coli and P. chrysogenum at k = 31, and a genome as MULTILINE FASTA for the streaming query.

What is checked is the reference's own contract (test/check_from_file.hpp:38-155): stream the build input -- every k-mer of the
file, in file order, every other one reverse-complemented -- and the ids must be 0, 1, 2, ... with the right orientation and
string fields; plus the CPU oracle on mixed batches, and for the streaming query the oracle's state machine and a brute-force
count of the positives (test/check.cpp:61-98)."""
from __future__ import annotations

import gzip
import os

import numpy as np
import pytest

import sshash_amd
from conftest import GOLDEN, has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]

CASES = [
    ("se.ust.k47.fa.gz", 47, 21, False),
    ("se.ust.k47.fa.gz", 47, 23, True),
    ("se.ust.k63.fa.gz", 63, 25, False),
    ("se.ust.k63.fa.gz", 63, 31, True),
    ("ecoli1_k31_ust.fa.gz", 31, 17, False),
    ("penicillium_chrysogenum_k31_ust.fa.gz", 31, 15, True),
]


def _rc64(v):
    v = v ^ np.uint64(0xAAAAAAAAAAAAAAAA)
    v = v.byteswap()
    v = ((v & np.uint64(0x0F0F0F0F0F0F0F0F)) << np.uint64(4)) | ((v >> np.uint64(4)) & np.uint64(0x0F0F0F0F0F0F0F0F))
    return ((v & np.uint64(0x3333333333333333)) << np.uint64(2)) | ((v >> np.uint64(2)) & np.uint64(0x3333333333333333))


def revcomp_packed(q: np.ndarray, k: int) -> np.ndarray:
    """(n, W) packed k-mers -> their reverse complements (reference include/kmer.hpp:159-165), vectorised for W = 1 and 2"""
    if q.shape[1] == 1:
        return (_rc64(q[:, 0]) >> np.uint64(64 - 2 * k))[:, None]
    r_hi, r_lo = _rc64(q[:, 0]), _rc64(q[:, 1])  # the words swap (kmer.hpp:162)
    s = np.uint64(128 - 2 * k)
    out = np.empty_like(q)
    out[:, 0] = (r_lo >> s) | (r_hi << (np.uint64(64) - s))
    out[:, 1] = r_hi >> s
    return out


def file_order_kmers(path: str, k: int):
    """every k-mer of the file in file order as (n, W) packed words, and per k-mer: string id, position in its string, length of its string"""
    from oracle.ground_truth import encode_bases, pack_kmers, read_fasta_sequences

    seqs = read_fasta_sequences(path, k)
    W = 1 if k <= 31 else 2
    los, his, sid, pos, lens = [], [], [], [], []
    for i, s in enumerate(seqs):
        lo, hi = pack_kmers(encode_bases(s), k)
        los.append(lo)
        his.append(hi)
        sid.append(np.full(lo.size, i, dtype=np.uint64))
        pos.append(np.arange(lo.size, dtype=np.uint64))
        lens.append(len(s))
    q = np.stack([np.concatenate(los), np.concatenate(his)], axis=1)[:, :W]
    return seqs, np.ascontiguousarray(q), np.concatenate(sid), np.concatenate(pos), np.array(lens, dtype=np.uint64)


@pytest.mark.parametrize("name,k,m,canonical", CASES, ids=[f"{c[0].split('.fa')[0]}-k{c[1]}-m{c[2]}-{'canonical' if c[3] else 'regular'}" for c in CASES])
def test_every_kmer_of_the_reference_file_in_file_order(name, k, m, canonical, tmp_path):
    from oracle import oracle as O

    path = os.path.join(GOLDEN, name)
    d = sshash_amd.Dictionary.build(path, k=k, m=m, canonical=canonical, num_threads=0).to_device(0)
    seqs, q, sid, pos, lens = file_order_kmers(path, k)
    n = q.shape[0]
    assert d.num_kmers() == n and d.num_strings() == len(seqs)
    assert d.device_stats()["sk_slots"] > 0
    # every other k-mer reverse-complemented (check_from_file.hpp:59-63)
    flipped = np.arange(n) % 2 == 0
    q2 = q.copy()
    q2[flipped] = revcomp_packed(q[flipped], k)
    got = d.lookup(np.ascontiguousarray(q2).reshape(-1), full=True)
    assert (got.kmer_id == np.arange(n, dtype=np.uint64)).all(), "wrong id assigned (test/check_from_file.hpp:66-72)"
    assert (got.kmer_orientation[flipped] == -1).all() and (got.kmer_orientation[~flipped] == 1).all()  # :79-83
    begin = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    assert (got.string_id == sid).all() and (got.kmer_id_in_string == pos).all()  # :92-133
    assert (got.string_begin == begin[sid.astype(np.int64)]).all() and (got.string_end == begin[sid.astype(np.int64) + 1]).all()
    assert (got.kmer_offset == begin[sid.astype(np.int64)] + pos).all()
    assert (got.minimizer_found == 1).all()
    assert (d.is_member(np.ascontiguousarray(q2).reshape(-1)) == 1).all()  # :157-162
    # access round trip (:146-155) on a sample, through the device
    ids = np.random.default_rng(k).integers(0, n, 50000)
    assert (d.access_packed(ids.astype(np.uint64)).reshape(-1, q.shape[1]) == q[ids]).all()
    # ASCII entry, lower-cased sequences (:38-44)
    take = np.random.default_rng(m).integers(0, len(seqs), 40)
    ascii_kmers, want = [], []
    first_id = np.concatenate([[0], np.cumsum(lens - np.uint64(k - 1))]).astype(np.int64)
    for s_id in take:
        s = seqs[int(s_id)].lower()
        for i in range(0, len(s) - k + 1, max(1, (len(s) - k + 1) // 50)):
            ascii_kmers.append(s[i:i + k])
            want.append(first_id[int(s_id)] + i)
    assert (d.lookup(ascii_kmers).kmer_id == np.array(want, dtype=np.uint64)).all()
    # the oracle on a mixed batch: positives on either strand, random negatives
    index = str(tmp_path / "ref.sshash")
    d.save(index)
    ora = O.OracleIndex(index)
    rng = np.random.default_rng(7)
    pick = rng.integers(0, n, 60000)
    mixed = np.concatenate([q2[pick], rng.integers(0, 1 << 62, (60000, q.shape[1]), dtype=np.uint64)])
    if q.shape[1] == 2:
        mixed[:, 1] &= np.uint64((1 << (2 * k - 64)) - 1)
    else:
        mixed[:, 0] &= np.uint64((1 << (2 * k)) - 1)
    mixed = np.ascontiguousarray(mixed[rng.permutation(mixed.shape[0])]).reshape(-1)
    full, want_o = d.lookup(mixed, full=True), ora.lookup_packed(mixed, True)
    for f in ("kmer_id", "kmer_id_in_string", "kmer_offset", "string_id", "string_begin", "string_end", "minimizer_found"):
        assert (getattr(full, f) == want_o[f]).all(), f
    assert (full.kmer_orientation.astype(np.int64) == want_o["kmer_orientation"]).all()
    d.close()


def multiline_segments(path: str):
    """src/query.cpp:9-47 + include/util.hpp:287-340: every line (headers included) is concatenated until an empty line ends
    the segment; the end of the file ends the last one."""
    data = gzip.open(path, "rb").read()
    lines = data.split(b"\n")
    if lines and lines[-1] == b"":
        lines = lines[:-1]
    segs, cur = [], []
    for line in lines:
        if line == b"":
            segs.append(b"".join(cur))
            cur = []
        else:
            cur.append(line)
    segs.append(b"".join(cur))
    return segs


@pytest.mark.parametrize("canonical", [False, True], ids=["regular", "canonical"])
def test_multiline_genome_streams_like_the_reference(canonical, tmp_path):
    """`sshash query --multiline` over data/queries/salmonella_enterica.fasta.gz against the salmonella dictionary: the six
    counters equal the oracle's state machine over the same segments, num_positive equals a brute-force membership count
    (test/check.cpp:61-98), and the identities of src/query.cpp:44-45 / streaming_query.hpp:113 hold."""
    from oracle import oracle as O
    from oracle.ground_truth import encode_bases, pack_kmers

    k, m = 31, 15
    path = os.path.join(GOLDEN, "salmonella_enterica.fasta.gz")
    d = sshash_amd.Dictionary.build(os.path.join(GOLDEN, "salmonella_enterica_k31_ust.fa.gz"), k=k, m=m, canonical=canonical, num_threads=0).to_device(0)
    rep = d.streaming_query_from_file(path, multiline=True)
    segs = [s for s in multiline_segments(path)]
    assert rep.num_kmers == sum(max(0, len(s) - k + 1) for s in segs)
    assert rep.num_kmers == rep.num_positive_kmers + rep.num_negative_kmers + rep.num_invalid_kmers
    assert rep.num_positive_kmers == rep.num_searches + rep.num_extensions
    index = str(tmp_path / "se.sshash")
    d.save(index)
    want = O.OracleIndex(index).streaming_query(segs)
    for f, v in want.items():
        assert int(getattr(rep, f)) == v, f
    # brute force: the dictionary's k-mers as a sorted array of canonical values; valid k-mers of the segments looked up in it
    _, q, _, _, _ = file_order_kmers(os.path.join(GOLDEN, "salmonella_enterica_k31_ust.fa.gz"), k)
    canon = np.sort(np.minimum(q[:, 0], revcomp_packed(q, k)[:, 0]))
    positives = invalid = 0
    for s in segs:
        if len(s) < k:
            continue
        a = np.frombuffer(s, dtype=np.uint8)
        ok = np.isin(a, np.frombuffer(b"ACGTacgt", dtype=np.uint8))
        bad = np.concatenate([[0], np.cumsum(~ok)])
        valid = (bad[k:] - bad[:-k]) == 0  # no invalid character among the k
        lo, _ = pack_kmers(encode_bases(s.decode("latin1")), k)
        c = np.minimum(lo, revcomp_packed(lo[:, None], k)[:, 0])[valid]
        at = np.minimum(np.searchsorted(canon, c), canon.size - 1)
        positives += int((canon[at] == c).sum())
        invalid += int((~valid).sum())
    assert rep.num_positive_kmers == positives and rep.num_invalid_kmers == invalid
    assert positives > 0.9 * rep.num_kmers  # the genome the unitigs were made from
    # per-k-mer results of the batched streaming lookup on a stretch of the first segment equal the point lookups (streaming_query.hpp:107)
    stretch = segs[0][:20000]
    results, rep2 = d.streaming_lookup([stretch])
    ids = results[0].kmer_id
    a = np.frombuffer(stretch, dtype=np.uint8)
    ok = np.isin(a, np.frombuffer(b"ACGTacgt", dtype=np.uint8))
    bad = np.concatenate([[0], np.cumsum(~ok)])
    valid = (bad[k:] - bad[:-k]) == 0
    lo, _ = pack_kmers(encode_bases(stretch.decode("latin1")), k)
    point = d.lookup(np.ascontiguousarray(lo[valid])).kmer_id
    assert (ids[valid] == point).all() and (ids[~valid] == sshash_amd.INVALID_U64).all()
    d.close()
