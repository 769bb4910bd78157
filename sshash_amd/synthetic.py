"""Synthetic spectrum-preserving string sets with the size statistics of the reference's datasets.

The reference's benchmark collections (S. enterica pangenome, human, ...) live on Zenodo and are
not available offline, so bench.py indexes a synthetic stand-in of the same scale instead
(16.4 M strings, 1.39 G bases, 894 M k-mers, k=31, m=21).
Three ingredients give the bucket-size skew real collections have:

  * background: uniformly random strings (min length k, geometric tail) -> singleton buckets;
  * SNP siblings: a short string (k <= len <= 2k-1) copied with ONE base changed at a position
    covered by all of its k-mers. Every k-mer differs, every m-mer not covering the base is
    shared -> buckets of 2..4 minimizer positions, as variant bubbles of a pangenome graph do;
  * hot motifs: low-hash m-mers planted in many strings as  D + motif + D  where D is the base-4
    spelling of the copy number (10 bases): the motif wins the minimizer election of every window
    containing it and all those k-mers stay distinct -> MIDLOAD and HEAVYLOAD buckets with a
    Zipf-like size distribution.

Everything is generated on packed arrays with numpy; output is what sshash_build_from_packed takes.
"""
from __future__ import annotations

import numpy as np

_MUL = np.uint64(0x517CC1B727220A95)


def _xxh64_u64(value: int, seed: int = 0) -> int:
    """XXH64 of one little-endian u64 (published xxHash algorithm) -- the m-mer hash magic."""
    M = (1 << 64) - 1
    P1, P2, P3, P4, P5 = (0x9E3779B185EBCA87, 0xC2B2AE3D27D4EB4F, 0x165667B19E3779F9, 0x85EBCA77C2B2AE63,
                          0x27D4EB2F165667C5)
    rotl = lambda v, r: ((v << r) | (v >> (64 - r))) & M
    h = (seed + P5 + 8) & M
    k1 = (rotl((value * P2) & M, 31) * P1) & M
    h ^= k1
    h = (rotl(h, 27) * P1 + P4) & M
    h ^= h >> 33
    h = (h * P2) & M
    h ^= h >> 29
    h = (h * P3) & M
    h ^= h >> 32
    return h


def pack_codes(codes: np.ndarray) -> np.ndarray:
    """uint8 base codes (0..3) -> 2-bit packed uint64 words, base i in bits [2i, 2i+1] of word i//32."""
    n = codes.size
    pad = (-n) % 32
    if pad:
        codes = np.concatenate([codes, np.zeros(pad, dtype=np.uint8)])
    q = codes.reshape(-1, 4)
    b = q[:, 0] | (q[:, 1] << 2) | (q[:, 2] << 4) | (q[:, 3] << 6)  # little-endian: byte j = bases 4j..4j+3
    return np.ascontiguousarray(b, dtype=np.uint8).view("<u8").copy()


def make_spss(num_bases: int, k: int = 31, m: int = 21, mean_len: float = 85.0, seed: int = 0x5555AAAA,
              sibling_fraction: float = 0.90, num_motifs: int = 2000, motif_copy_fraction: float = 0.04,
              max_motif_copies: int = 40000, build_seed: int = 1):
    """-> (packed words, endpoints). Deterministic for a given argument tuple."""
    rng = np.random.default_rng(seed)
    # ---- string lengths -------------------------------------------------------------------
    est = int(num_bases / mean_len * 1.05) + 16
    lens = k + rng.geometric(1.0 / max(1.0, mean_len - k + 1), est) - 1
    lens = lens.astype(np.int64)
    csum = np.cumsum(lens)
    n_str = int(np.searchsorted(csum, num_bases)) + 1
    lens = lens[:n_str]
    # sibling groups: string i+1 (and sometimes i+2) repeats string i with one base changed
    cand = np.arange(max(0, n_str - 2))
    pick = cand[rng.random(cand.size) < sibling_fraction / 2.3]
    pick = pick[np.concatenate([[True], np.diff(pick) > 2])] if pick.size else pick  # keep groups disjoint
    triple = rng.random(pick.size) < 0.3
    lens[pick + 1] = lens[pick]
    lens[pick[triple] + 2] = lens[pick[triple]]
    endpoints = np.zeros(n_str + 1, dtype=np.uint64)
    endpoints[1:] = np.cumsum(lens).astype(np.uint64)
    total = int(endpoints[-1])
    codes = rng.integers(0, 4, total, dtype=np.uint8)
    begin = endpoints[:-1].astype(np.int64)

    def copy_with_snp(src, dst, second):
        """dst := src with one base changed every 30 positions (so every k-mer window, k >= 31, holds a
        changed base and differs from its twin, while the m-mers between two changes are shared);
        `second` applies another substitution at the SAME sites so that the three strings of a group
        are pairwise different."""
        spacing = k - 1
        for L in np.unique(lens[src]):
            sel = lens[src] == L
            s, d = begin[src[sel]], begin[dst[sel]]
            idx = np.arange(L, dtype=np.int64)
            block = codes[s[:, None] + idx[None, :]]
            key = block[:, :8].astype(np.int64) @ (4 ** np.arange(8, dtype=np.int64))  # content-derived
            if L <= 2 * k - 1:
                p = (L - k) + key % (2 * k - L)  # one site covered by every window
            else:
                p = key % spacing
            first_delta = (1 + (key // 64) % 2).astype(np.uint8)  # 1 or 2
            delta = np.where(first_delta == 1, np.uint8(2), np.uint8(3)).astype(np.uint8) if second else first_delta
            rel = idx[None, :] - p[:, None]
            mask = (rel >= 0) & (rel % spacing == 0)
            block = np.where(mask, (block + delta[:, None]) & 3, block).astype(np.uint8)
            codes[d[:, None] + idx[None, :]] = block

    if pick.size:
        copy_with_snp(pick, pick + 1, False)
        if triple.any():
            copy_with_snp(pick[triple], pick[triple] + 2, True)

    # ---- hot motifs ---------------------------------------------------------------------------
    used = np.zeros(n_str, dtype=bool)
    if pick.size:
        used[pick] = used[pick + 1] = True
        used[pick[triple] + 2] = True
    flank = 10
    need = m + 2 * flank
    hosts = np.nonzero((lens >= need + 2) & ~used)[0]
    n_copies_total = int(min(hosts.size, motif_copy_fraction * n_str))
    if num_motifs > 0 and n_copies_total > 0:
        magic = np.uint64(_xxh64_u64(build_seed, 0))
        n_try = max(1 << 22, 4000 * num_motifs)
        cand_m = rng.integers(0, 1 << (2 * m), n_try, dtype=np.uint64)
        with np.errstate(over="ignore"):
            hashes = (cand_m * _MUL) ^ magic
        best = np.argsort(hashes)[: num_motifs * 2]
        motifs = np.unique(cand_m[best])[:num_motifs]
        rng.shuffle(motifs)
        # Zipf-like copy counts, at least 2, capped, rescaled to the budget
        w = 1.0 / np.arange(1, motifs.size + 1) ** 0.9
        counts = np.maximum(2, (w / w.sum() * n_copies_total).astype(np.int64))
        counts = np.minimum(counts, min(max_motif_copies, 4 ** flank - 1))
        while counts.sum() > n_copies_total and counts.max() > 2:
            counts = np.maximum(2, (counts * 0.95).astype(np.int64))
        n_used = int(min(counts.sum(), hosts.size))
        host_ids = rng.choice(hosts, n_used, replace=False)
        motif_of = np.repeat(np.arange(motifs.size), counts)[:n_used]
        copy_no = (np.arange(counts.sum()) - np.repeat(np.cumsum(counts) - counts, counts))[:n_used]
        # segment = D + motif + D
        seg = np.empty((n_used, need), dtype=np.uint8)
        digits = np.stack([(copy_no >> (2 * (flank - 1 - j))) & 3 for j in range(flank)], axis=1).astype(np.uint8)
        mot = np.stack([(motifs[motif_of] >> np.uint64(2 * j)) & np.uint64(3) for j in range(m)], axis=1).astype(np.uint8)
        seg[:, :flank] = digits
        seg[:, flank:flank + m] = mot
        seg[:, flank + m:] = digits
        at = begin[host_ids] + (lens[host_ids] - need) // 2
        codes[at[:, None] + np.arange(need)[None, :]] = seg
    words = pack_codes(codes)
    return words, endpoints


def draw_queries(d, n: int, positive_fraction: float = 0.5, seed: int = 0x5555AAAA) -> np.ndarray:
    """The reference's benchmark mix (tools/perf.hpp:38-51,67-74): positives = access(random id) with
    every other one reverse-complemented, negatives = uniformly random k-mers; shuffled; seeded."""
    rng = np.random.default_rng(seed)
    W, k = d.words_per_kmer(), d.k()
    n_pos = int(n * positive_fraction)
    ids = rng.integers(0, d.num_kmers(), n_pos, dtype=np.uint64)
    pos = d.access_packed(ids).reshape(n_pos, W)
    if W == 1:
        x = pos[::2, 0] ^ np.uint64(0xAAAAAAAAAAAAAAAA)
        x = x.byteswap()
        x = ((x & np.uint64(0x0F0F0F0F0F0F0F0F)) << np.uint64(4)) | ((x >> np.uint64(4)) & np.uint64(0x0F0F0F0F0F0F0F0F))
        x = ((x & np.uint64(0x3333333333333333)) << np.uint64(2)) | ((x >> np.uint64(2)) & np.uint64(0x3333333333333333))
        pos[::2, 0] = x >> np.uint64(64 - 2 * k)
    else:
        def rc64(v):
            v = v ^ np.uint64(0xAAAAAAAAAAAAAAAA)
            v = v.byteswap()
            v = ((v & np.uint64(0x0F0F0F0F0F0F0F0F)) << np.uint64(4)) | ((v >> np.uint64(4)) & np.uint64(0x0F0F0F0F0F0F0F0F))
            return ((v & np.uint64(0x3333333333333333)) << np.uint64(2)) | ((v >> np.uint64(2)) & np.uint64(0x3333333333333333))
        lo, hi = pos[::2, 0].copy(), pos[::2, 1].copy()
        r_hi, r_lo = rc64(lo), rc64(hi)  # word order swaps (reference include/kmer.hpp:162)
        s = np.uint64(128 - 2 * k)       # 2 <= s < 64 for 33 <= k <= 63
        pos[::2, 0] = (r_lo >> s) | (r_hi << (np.uint64(64) - s))
        pos[::2, 1] = r_hi >> s
    neg = rng.integers(0, 1 << 63, (n - n_pos, W), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, (n - n_pos, W), dtype=np.uint64)
    if W == 1:
        neg[:, 0] &= np.uint64((1 << (2 * k)) - 1)
    else:
        neg[:, 1] &= np.uint64((1 << (2 * k - 64)) - 1)
    allq = np.concatenate([pos, neg])
    allq = allq[rng.permutation(n)]  # (a row shuffle of an (n,1) array is an order of magnitude slower)
    return np.ascontiguousarray(allq).reshape(-1)


# ---- the same query mix, drawn on the GPU (10^9-query batches: BASELINE.json configs[2]) ---------------------

def _s64(v: int) -> int:
    """Python int (unsigned 64-bit pattern) -> the int64 value with the same bits."""
    return v - (1 << 64) if v >= (1 << 63) else v


def _shr(x, s: int):
    """Logical right shift of an int64 tensor (torch's >> is arithmetic)."""
    return (x >> s) & ((1 << (64 - s)) - 1) if s else x


def _reverse_pairs64(x):
    """Reverse the order of the 32 two-bit groups of every int64 (torch has no byteswap)."""
    x = ((x & 0x3333333333333333) << 2) | (_shr(x, 2) & 0x3333333333333333)
    x = ((x & 0x0F0F0F0F0F0F0F0F) << 4) | (_shr(x, 4) & 0x0F0F0F0F0F0F0F0F)
    x = ((x & 0x00FF00FF00FF00FF) << 8) | (_shr(x, 8) & 0x00FF00FF00FF00FF)
    x = ((x & 0x0000FFFF0000FFFF) << 16) | (_shr(x, 16) & 0x0000FFFF0000FFFF)
    return (x << 32) | _shr(x, 32)


def revcomp_device(q, k: int):
    """Reverse complement of packed k-mers held in an int64 torch tensor of shape (n, W)
    (reference include/kmer.hpp:159-165: complement = x ^ 0xAAAA..., reverse the 2-bit groups, shift down)."""
    comp = _s64(0xAAAAAAAAAAAAAAAA)
    if q.shape[1] == 1:
        return _shr(_reverse_pairs64(q[:, 0] ^ comp), 64 - 2 * k).unsqueeze(1)
    import torch

    r_hi, r_lo = _reverse_pairs64(q[:, 0] ^ comp), _reverse_pairs64(q[:, 1] ^ comp)  # the words swap (kmer.hpp:162)
    s = 128 - 2 * k  # 2 <= s <= 62 for 33 <= k <= 63
    return torch.stack([_shr(r_lo, s) | (r_hi << (64 - s)), _shr(r_hi, s)], dim=1)


def draw_queries_device(d, device: int, n: int, positive_fraction: float = 0.5, seed: int = 0x5555AAAA,
                        negatives: str = "random", chunk: int = 1 << 26):
    """draw_queries() on the GPU: -> int64 torch tensor of n*W packed words on cuda:`device`.

    positives = access(random id) on the device (sshash_access_packed_device), every other one (a fair coin)
    reverse-complemented; negatives = "random": uniformly random k-mers (tools/perf.hpp:67-74), or "mutated": a
    random indexed k-mer with ONE base substituted -- absent (unless the substitution lands on an indexed sibling)
    but sharing its minimizers with the index, the way reads of a related genome do. Positions are mixed by an
    independent coin per query, so no shuffle is needed. Deterministic for (seed, n, chunk)."""
    import torch

    dev = torch.device("cuda", device)
    W, k, nk = d.words_per_kmer(), d.k(), d.num_kmers()
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed) & 0x7FFFFFFFFFFFFFFF)
    out = torch.empty((n, W), dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def indexed(m):
        ids = torch.randint(0, nk, (m,), generator=g, device=dev, dtype=torch.int64)
        x = torch.empty((m, W), dtype=torch.int64, device=dev)
        d.access_packed_device(device, ids.data_ptr(), m, x.data_ptr(), stream=stream)
        return x

    def rand64(m):
        hi = torch.randint(0, 1 << 32, (m,), generator=g, device=dev, dtype=torch.int64)
        lo = torch.randint(0, 1 << 32, (m,), generator=g, device=dev, dtype=torch.int64)
        return (hi << 32) | lo

    for at in range(0, n, chunk):
        m = min(chunk, n - at)
        pos = indexed(m)
        flip = torch.rand(m, generator=g, device=dev) < 0.5
        pos = torch.where(flip.unsqueeze(1), revcomp_device(pos, k), pos)
        if negatives == "random":
            neg = torch.stack([rand64(m) for _ in range(W)], dim=1)
            if W == 1:
                neg[:, 0] &= (1 << (2 * k)) - 1
            else:
                neg[:, 1] &= (1 << (2 * k - 64)) - 1
        elif negatives == "mutated":
            neg = indexed(m)
            p = torch.randint(0, k, (m,), generator=g, device=dev, dtype=torch.int64)
            delta = torch.randint(1, 4, (m,), generator=g, device=dev, dtype=torch.int64)
            if W == 1:
                neg[:, 0] ^= delta << (2 * p)
            else:
                low = p < 32
                neg[:, 0] ^= torch.where(low, delta << (2 * (p & 31)), torch.zeros_like(delta))
                neg[:, 1] ^= torch.where(low, torch.zeros_like(delta), delta << (2 * (p & 31)))
            flip = torch.rand(m, generator=g, device=dev) < 0.5
            neg = torch.where(flip.unsqueeze(1), revcomp_device(neg, k), neg)
        else:
            raise ValueError("negatives must be 'random' or 'mutated'")
        is_pos = torch.rand(m, generator=g, device=dev) < positive_fraction
        out[at:at + m] = torch.where(is_pos.unsqueeze(1), pos, neg)
    return out.reshape(-1)


# ---- synthetic FASTQ of reads drawn from a dictionary (BASELINE.json configs[3]) -----------------

def make_reads_device(d, device: int, n_reads: int, read_len: int = 150, positive_fraction: float = 0.5, substitution_rate: float = 0.01,
                      n_rate: float = 1e-3, seed: int = 0x5555AAAA):
    """-> uint8 torch tensor (n_reads, read_len) of ASCII bases on cuda:`device`. A positive read spells read_len - k + 1
    consecutive k-mers of the dictionary (ids id .. id + read_len - k: a read that runs past the end of its string continues in
    the next one, as a chimeric read would), every other one reverse-complemented, then gets substitutions at `substitution_rate`
    per base; the other reads are uniformly random; 'N' replaces a base at `n_rate`."""
    import torch

    dev = torch.device("cuda", device)
    k, nk, W = d.k(), d.num_kmers(), d.words_per_kmer()
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed) & 0x7FFFFFFFFFFFFFFF)
    stream = torch.cuda.current_stream(dev).cuda_stream
    span = read_len - k + 1
    out = torch.empty((n_reads, read_len), dtype=torch.uint8, device=dev)
    chunk = max(1, (1 << 26) // (span * W))
    alphabet = torch.tensor(list(b"ACTG"), dtype=torch.uint8, device=dev)  # the reference's 2-bit codes (include/kmer.hpp:118)
    comp = torch.tensor([2, 3, 0, 1], dtype=torch.int64, device=dev)       # A<->T, C<->G as codes
    for at in range(0, n_reads, chunk):
        m = min(chunk, n_reads - at)
        first = torch.randint(0, max(1, nk - span), (m,), generator=g, device=dev, dtype=torch.int64)
        ids = (first[:, None] + torch.arange(span, device=dev, dtype=torch.int64)[None, :]).reshape(-1).contiguous()
        km = torch.empty((m * span, W), dtype=torch.int64, device=dev)
        d.access_packed_device(device, ids.data_ptr(), m * span, km.data_ptr(), stream=stream)
        km = km.reshape(m, span, W)
        codes = torch.empty((m, read_len), dtype=torch.int64, device=dev)
        codes[:, :span] = km[:, :, 0] & 3  # the first base of every k-mer of the run ...
        last = km[:, span - 1, :]          # ... and the other k - 1 bases of the last one (base j: bits 2j, 2j+1 of its words)
        for j in range(1, k):
            codes[:, span - 1 + j] = (last[:, j // 32] >> (2 * (j % 32))) & 3
        flip = torch.rand(m, generator=g, device=dev) < 0.5
        codes = torch.where(flip[:, None], comp[codes.flip(1)], codes)
        sub = torch.rand((m, read_len), generator=g, device=dev) < substitution_rate
        codes = torch.where(sub, (codes + torch.randint(1, 4, (m, read_len), generator=g, device=dev)) & 3, codes)
        rnd = torch.randint(0, 4, (m, read_len), generator=g, device=dev)
        positive = torch.rand(m, generator=g, device=dev) < positive_fraction
        codes = torch.where(positive[:, None], codes, rnd)
        text = alphabet[codes]
        text = torch.where(torch.rand((m, read_len), generator=g, device=dev) < n_rate, torch.full_like(text, ord("N")), text)
        out[at:at + m] = text
    return out


def bgzf_compress(raw: bytes, level: int = 6) -> bytes:
    """raw -> BGZF members (what bgzip / htslib write: gzip members of at most 64 KiB of input each, the member's size in the
    'BC' extra subfield of its header; SAM/BAM specification section 4.1), without the empty end-of-file member."""
    import struct
    import zlib

    out = []
    for at in range(0, len(raw), 65280):
        piece = raw[at:at + 65280]
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = c.compress(piece) + c.flush()
        size = 18 + len(body) + 8
        if size > 65536:
            raise ValueError("bgzf_compress: a block that does not compress")
        out.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", size - 1) + body +
                   struct.pack("<II", zlib.crc32(piece) & 0xFFFFFFFF, len(piece)))
    return b"".join(out)


BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def write_fastq(reads, path: str, gzip_level: int | None = None, workers: int = 8, bgzf: bool = False) -> int:
    """reads: uint8 array/tensor (n, L) of ASCII bases -> a FASTQ file (4 lines per read); gzip_level: also compress -- the
    file is written as independent gzip members by `workers` processes (a multi-member .gz is what `cat a.gz b.gz` makes;
    zlib's gzread, the reference's zip_istream and gunzip all read it as one stream); bgzf: as BGZF members (bgzip's format,
    which the file reader inflates on all cores). Returns the bytes written."""
    import os
    import zlib
    from concurrent.futures import ThreadPoolExecutor

    a = reads.cpu().numpy() if hasattr(reads, "cpu") else np.asarray(reads)
    n, L = a.shape

    def records(lo, hi):
        m = hi - lo
        idx = np.arange(lo, hi)
        rec = np.empty((m, 11 + 1 + L + 1 + 2 + L + 1), dtype=np.uint8)
        rec[:, 0] = ord("@")
        rec[:, 1] = ord("r")
        for j in range(9):  # nine decimal digits of the read's number
            rec[:, 2 + j] = ord("0") + (idx // 10 ** (8 - j)) % 10
        rec[:, 11] = 10
        rec[:, 12:12 + L] = a[lo:hi]
        rec[:, 12 + L] = 10
        rec[:, 13 + L] = ord("+")
        rec[:, 14 + L] = 10
        rec[:, 15 + L:15 + 2 * L] = ord("I")
        rec[:, 15 + 2 * L] = 10
        return rec.tobytes()

    step = 1 << 18
    pieces = [(lo, min(n, lo + step)) for lo in range(0, n, step)]

    def make(p):
        raw = records(*p)
        if gzip_level is None:
            return raw
        if bgzf:
            return bgzf_compress(raw, gzip_level)
        c = zlib.compressobj(gzip_level, zlib.DEFLATED, 31)  # 31: gzip container
        return c.compress(raw) + c.flush()

    total = 0
    with open(path, "wb") as f, ThreadPoolExecutor(max_workers=workers) as ex:  # (zlib and numpy release the GIL)
        for blob in ex.map(make, pieces):
            f.write(blob)
            total += len(blob)
        if bgzf and gzip_level is not None:
            f.write(BGZF_EOF)
            total += len(BGZF_EOF)
    return total
