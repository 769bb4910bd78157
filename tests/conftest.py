"""Shared fixtures. GPU tests are marked `gpu`; everything else runs on a CPU-only box."""
from __future__ import annotations

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
SE_FASTA = os.path.join(GOLDEN, "salmonella_enterica_k31_ust.fa.gz")
K63_FASTA = os.path.join(GOLDEN, "se.ust.k63.head.fa.gz")
WEIGHTED_FASTA = os.path.join(GOLDEN, "salmonella_enterica.weighted.ust.k31.fa.gz")  # reference data/unitigs_stitched/with_weights
FASTQ = os.path.join(GOLDEN, "SRR5833294.10K.fastq.gz")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a visible MI355X (run with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the shared libraries exist (the driver normally runs build() first)."""
    import sshash_amd

    if not os.path.exists(sshash_amd.library_path()):
        import __graft_entry__

        __graft_entry__.build()
    from oracle import oracle as O

    O.build()


def has_gpu() -> bool:
    import sshash_amd

    try:
        return sshash_amd.device_count() > 0
    except Exception:
        return False


class Case:
    """One dictionary + everything needed to check it."""

    def __init__(self, name, sequences, k, m, canonical, tmpdir, fasta=None):
        import sshash_amd
        from oracle import oracle as O
        from oracle.ground_truth import GroundTruth

        self.name, self.k, self.m, self.canonical = name, k, m, canonical
        self.sequences = sequences
        if fasta is None:
            fasta = os.path.join(tmpdir, name + ".fa")
            with open(fasta, "w") as f:
                for s in sequences:
                    f.write(">\n" + s + "\n")
        self.fasta = fasta
        self.dict = sshash_amd.Dictionary.build(fasta, k=k, m=m, canonical=canonical, num_threads=4)
        self.index_path = os.path.join(tmpdir, name + ".sshash")
        self.dict.save(self.index_path)
        self.oracle = O.OracleIndex(self.index_path)
        self.gt = GroundTruth(sequences, k)
        self.W = 1 if k <= 31 else 2

    def queries(self, n_pos, n_neg, seed=0):
        """perf.hpp-style batch: positives by random id (every other one reverse-complemented),
        negatives uniformly random; shuffled. Returns packed words (n*W)."""
        rng = np.random.default_rng(seed)
        ids = rng.integers(0, self.gt.num_kmers, n_pos)
        pos = self.gt.kmers(ids).reshape(n_pos, self.W)
        rc = self.gt._revcomp(pos[::2].reshape(-1)).reshape(-1, self.W)
        pos[::2] = rc
        neg = rng.integers(0, 1 << 62, (n_neg, self.W), dtype=np.uint64)
        if self.W == 2:
            neg[:, 1] &= np.uint64((1 << (2 * self.k - 64)) - 1)
        else:
            neg[:, 0] &= np.uint64((1 << (2 * self.k)) - 1)
        allq = np.concatenate([pos, neg])
        perm = rng.permutation(allq.shape[0])
        return np.ascontiguousarray(allq[perm]).reshape(-1)


_ALPHABET = "ACTG"  # code -> char, reference include/kmer.hpp:118


def random_dna(rng, n):
    return "".join(_ALPHABET[i] for i in rng.integers(0, 4, n))


def mmer_hash(mmer_str, seed=1):
    """(x * 0x517cc1b727220a95) ^ xxh64(seed) -- reference include/hash_util.hpp:88-91."""
    from oracle import oracle as O

    x = 0
    for i, c in enumerate(mmer_str):
        x |= ((ord(c) >> 1) & 3) << (2 * i)
    return ((x * 0x517CC1B727220A95) & ((1 << 64) - 1)) ^ O.xxh64_u64(seed, 0)


def table_key_hash(mmer_str, k=31):
    """sk_select_hash of csrc/device_layout.hpp (the super-k-mer table's own election hash: 26 bits over the first
    min(m, 16) bases of an m-mer occurrence), taken over both strands: an m-mer with a small value wins the table's
    key election of any window it appears in."""
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}

    def h(t):
        x = 0
        for i, c in enumerate(t[:16]):
            x |= ((ord(c) >> 1) & 3) << (2 * i)
        mask = 0xFFFFFFFF if len(t) >= 16 else (1 << (2 * len(t))) - 1
        salt = 0x6A09E667 if k <= 31 else 0xAAAAAAAA  # SK_SELECT_SALT / SK_SELECT_FLIP
        return ((((x ^ salt) & mask) * (0x9E3779B1 << 6)) & 0xFFFFFFFF) >> 6

    return min(h(mmer_str), h("".join(comp[c] for c in reversed(mmer_str))))


def skewed_sequences(k, m, seed=3, n_heavy=150, n_mid=7, n_plain=60, canonical=False):
    """Synthetic strings with planted low-hash m-mers so that MIDLOAD and HEAVYLOAD buckets exist.
    Every planted copy sits between fresh random flanks, so k-mers stay unique."""
    rng = np.random.default_rng(seed)
    # m-mers with the smallest hash among many candidates: they win the minimizer election of
    # (almost) any window they appear in
    cands = sorted({random_dna(rng, m) for _ in range(20000)}, key=mmer_hash)
    motifs = cands[:3]
    seqs = []
    for i in range(n_heavy):
        seqs.append(random_dna(rng, k + int(rng.integers(5, 40))) + motifs[0] + random_dna(rng, k + int(rng.integers(5, 40))))
    for i in range(n_mid):
        seqs.append(random_dna(rng, k + 10) + motifs[1] + random_dna(rng, k + 25))
    for i in range(2):
        seqs.append(random_dna(rng, k + 3) + motifs[2] + random_dna(rng, 2 * k))
    for i in range(n_plain):
        seqs.append(random_dna(rng, int(rng.integers(k, 6 * k))))
    seqs.append(random_dna(rng, k))  # a string holding exactly one k-mer
    # the same for the device's super-k-mer table, which elects keys with its own hash: one key with more
    # occurrences than the table lists (left to the complete path), one with a list in `occ`, one with two
    table_motifs = [c for c in sorted(cands, key=lambda c: table_key_hash(c, k)) if c not in motifs][:3]
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}

    def canonical_kmers(t):
        out = set()
        for i in range(len(t) - k + 1):
            x = t[i:i + k]
            out.add(min(x, "".join(comp[c] for c in reversed(x))))
        return out

    seen = set()
    for t in seqs:
        seen |= canonical_kmers(t)
    for motif, copies in zip(table_motifs, (100, 9, 2)):
        for i in range(copies):
            while True:  # short k: keep every k-mer of the collection distinct
                t = random_dna(rng, k + int(rng.integers(3, 30))) + motif + random_dna(rng, k + int(rng.integers(3, 30)))
                mine = canonical_kmers(t)
                if len(mine) == len(t) - k + 1 and not (mine & seen):
                    break
            seen |= mine
            seqs.append(t)
    return seqs


@pytest.fixture(scope="session")
def tmp_session(tmp_path_factory):
    return str(tmp_path_factory.mktemp("sshash"))


@pytest.fixture(scope="session")
def se_sequences():
    from oracle.ground_truth import read_fasta_sequences

    return read_fasta_sequences(SE_FASTA, 31)


@pytest.fixture(scope="session")
def case_se_regular(tmp_session, se_sequences):
    return Case("se_regular", se_sequences, 31, 13, False, tmp_session, fasta=SE_FASTA)


@pytest.fixture(scope="session")
def case_se_canonical(tmp_session, se_sequences):
    return Case("se_canonical", se_sequences, 31, 13, True, tmp_session, fasta=SE_FASTA)


@pytest.fixture(scope="session")
def case_skew_regular(tmp_session):
    return Case("skew_regular", skewed_sequences(31, 11), 31, 11, False, tmp_session)


@pytest.fixture(scope="session")
def case_skew_canonical(tmp_session):
    return Case("skew_canonical", skewed_sequences(31, 11, seed=5), 31, 11, True, tmp_session)


@pytest.fixture(scope="session")
def case_k63_regular(tmp_session):
    from oracle.ground_truth import read_fasta_sequences

    return Case("k63_regular", read_fasta_sequences(K63_FASTA, 63), 63, 25, False, tmp_session, fasta=K63_FASTA)


@pytest.fixture(scope="session")
def case_k63_canonical(tmp_session):
    return Case("k63_canonical", skewed_sequences(63, 17, seed=11, n_heavy=90, n_plain=30), 63, 17, True, tmp_session)


@pytest.fixture(scope="session")
def case_small_k(tmp_session):
    return Case("small_k", skewed_sequences(15, 7, seed=13, n_heavy=80, n_plain=40), 15, 7, False, tmp_session)


@pytest.fixture(scope="session")
def case_m_equals_k(tmp_session):
    """m == k: one m-mer per k-mer, the minimizer IS the k-mer (every bucket a singleton)."""
    rng = np.random.default_rng(21)
    seqs = [random_dna(rng, int(rng.integers(21, 200))) for _ in range(150)] + [random_dna(rng, 21)]
    return Case("m_equals_k", seqs, 21, 21, True, tmp_session)


ALL_SMALL_CASES = ["case_skew_regular", "case_skew_canonical", "case_k63_canonical", "case_small_k", "case_m_equals_k"]
