"""Independent ground truth for SSHash lookups, built straight from the input sequences.

TEST INFRASTRUCTURE ONLY (same usage rule as the rest of oracle/).

k-mer ids are defined by input order (reference include/spectrum_preserving_string_set.hpp:227,
asserted by test/check_from_file.hpp:66-72): streaming the input file, the i-th k-mer has id i.
So a plain table {k-mer -> rank in the file} pins every field of lookup_result without running
any index code at all:

    kmer_id            rank of the k-mer in the file
    string_id          index of the sequence it lies in
    kmer_id_in_string  position inside that sequence
    kmer_offset        base offset in the concatenation of all sequences
    string_begin/end   base offsets of that sequence in the concatenation
    kmer_orientation   +1 if the query equals the stored k-mer, -1 if its reverse complement does
                       (regular index: forward is tried first, src/dictionary.cpp:70-76;
                        canonical index: the stored strand decides, spss.hpp:261-264)
"""
from __future__ import annotations

import gzip
from typing import List, Sequence

import numpy as np

INVALID = np.uint64(0xFFFFFFFFFFFFFFFF)
_M64 = (1 << 64) - 1


def read_fasta_sequences(path: str, k: int) -> List[str]:
    """Header line + ONE sequence line per record; a final line without newline is dropped
    (reference src/builder/encode_strings.cpp:136-140)."""
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rb") as f:
        data = f.read()
    lines = data.split(b"\n")
    complete = lines[:-1]  # the piece after the last '\n' is not a complete line
    seqs = [complete[i].decode("ascii") for i in range(1, len(complete), 2)]
    for s in seqs:
        if len(s) < k:
            raise ValueError("sequence shorter than k")
    return seqs


def encode_bases(s: str) -> np.ndarray:
    a = np.frombuffer(s.encode("ascii"), dtype=np.uint8)
    return ((a >> 1) & 3).astype(np.uint64)  # A0 C1 T2 G3, reference include/kmer.hpp:194


def pack_kmers(codes: np.ndarray, k: int):
    """All k-mers of a code array -> (lo, hi) uint64 arrays, first base in the low bits."""
    n = codes.size - k + 1
    lo = np.zeros(n, dtype=np.uint64)
    hi = np.zeros(n, dtype=np.uint64)
    for j in range(k):
        c = codes[j:j + n]
        if j < 32:
            lo |= c << np.uint64(2 * j)
        else:
            hi |= c << np.uint64(2 * (j - 32))
    return lo, hi


def revcomp_int(x: int, k: int) -> int:
    r = 0
    for _ in range(k):
        r = (r << 2) | ((x & 3) ^ 2)
        x >>= 2
    return r


def _revcomp_u64(x: np.ndarray, k: int) -> np.ndarray:
    """Vectorised reverse complement of k-mers held in one uint64 (k <= 32)."""
    x = x ^ np.uint64(0xAAAAAAAAAAAAAAAA)
    x = x.byteswap()
    x = ((x & np.uint64(0x0F0F0F0F0F0F0F0F)) << np.uint64(4)) | ((x >> np.uint64(4)) & np.uint64(0x0F0F0F0F0F0F0F0F))
    x = ((x & np.uint64(0x3333333333333333)) << np.uint64(2)) | ((x >> np.uint64(2)) & np.uint64(0x3333333333333333))
    return x >> np.uint64(64 - 2 * k)


class GroundTruth:
    def __init__(self, sequences: Sequence[str], k: int):
        self.k = k
        self.W = 1 if k <= 31 else 2
        lens = np.array([len(s) for s in sequences], dtype=np.uint64)
        self.endpoints = np.concatenate([[np.uint64(0)], np.cumsum(lens, dtype=np.uint64)])
        self.num_strings = len(sequences)
        self.num_kmers = int((lens - np.uint64(k - 1)).sum())
        lo_parts, hi_parts, sid_parts, pos_parts = [], [], [], []
        for s_id, s in enumerate(sequences):
            lo, hi = pack_kmers(encode_bases(s), k)
            lo_parts.append(lo)
            hi_parts.append(hi)
            sid_parts.append(np.full(lo.size, s_id, dtype=np.uint64))
            pos_parts.append(np.arange(lo.size, dtype=np.uint64))
        self.lo = np.concatenate(lo_parts)
        self.hi = np.concatenate(hi_parts)
        self.string_id = np.concatenate(sid_parts)
        self.in_string = np.concatenate(pos_parts)
        if self.W == 1:
            self._order = np.argsort(self.lo, kind="stable")
            self._sorted = self.lo[self._order]
        else:
            self._table = {}
            for i, (l, h) in enumerate(zip(self.lo.tolist(), self.hi.tolist())):
                self._table.setdefault((h << 64) | l, i)

    # -- forward k-mers by id ------------------------------------------------------------------
    def kmers(self, ids: np.ndarray) -> np.ndarray:
        ids = np.asarray(ids, dtype=np.int64)
        if self.W == 1:
            return self.lo[ids].copy()
        out = np.empty((ids.size, 2), dtype=np.uint64)
        out[:, 0] = self.lo[ids]
        out[:, 1] = self.hi[ids]
        return out.reshape(-1)

    def _find_exact(self, q: np.ndarray) -> np.ndarray:
        """-> index of each packed query among the stored (forward) k-mers, -1 if absent."""
        if self.W == 1:
            at = np.searchsorted(self._sorted, q)
            at_c = np.minimum(at, self._sorted.size - 1)
            hit = self._sorted[at_c] == q
            return np.where(hit, self._order[at_c], -1).astype(np.int64)
        q2 = q.reshape(-1, 2)
        return np.array([self._table.get((int(h) << 64) | int(l), -1) for l, h in q2], dtype=np.int64)

    def _revcomp(self, q: np.ndarray) -> np.ndarray:
        if self.W == 1:
            return _revcomp_u64(q, self.k)
        q2 = q.reshape(-1, 2)
        out = np.empty_like(q2)
        for i, (l, h) in enumerate(q2):
            r = revcomp_int((int(h) << 64) | int(l), self.k)
            out[i, 0] = r & _M64
            out[i, 1] = r >> 64
        return out.reshape(-1)

    def lookup(self, queries: np.ndarray, check_rc: bool = True) -> dict:
        q = np.ascontiguousarray(queries, dtype=np.uint64)
        fwd = self._find_exact(q)
        idx = fwd.copy()
        ori = np.ones(idx.size, dtype=np.int8)
        if check_rc:
            need = fwd < 0
            if need.any():
                if self.W == 1:
                    bwd = self._find_exact(self._revcomp(q[need]))
                else:
                    bwd = self._find_exact(self._revcomp(q.reshape(-1, 2)[need].reshape(-1)))
                idx[need] = bwd
                ori[need] = -1
        found = idx >= 0
        safe = np.where(found, idx, 0)
        k1 = np.uint64(self.k - 1)
        sid = self.string_id[safe]
        begin = self.endpoints[sid.astype(np.int64)]
        end = self.endpoints[sid.astype(np.int64) + 1]
        in_s = self.in_string[safe]

        def masked(a):
            return np.where(found, a, INVALID).astype(np.uint64)

        return {
            "found": found,
            "kmer_id": masked(safe.astype(np.uint64)),
            "kmer_id_in_string": masked(in_s),
            "kmer_offset": masked(safe.astype(np.uint64) + sid * k1),
            "string_id": masked(sid),
            "string_begin": masked(begin),
            "string_end": masked(end),
            "kmer_orientation": ori,
        }
