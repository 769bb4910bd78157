"""Synthetic spectrum-preserving string sets whose REPEAT STRUCTURE is a genome's, not a hash function's.

The reference's benchmark collections are not available offline, so bench.py indexes a stand-in. The stand-in of
rounds 1-2 (synthetic.make_spss) planted m-mers with a low *reference* minimizer hash: that reproduces bucket sizes for
the reference's own minimizer order and for no other -- any structure keyed differently (this repo's super-k-mer table
elects its key with its own hash) saw almost no repeats. Here repeats are made the way genomes make them, independent of
any hash order:

  * background   uniformly random strings -> singleton buckets;
  * families     a random consensus of `length` bases, `copies` copies, every copy with independent substitutions at
                 rate `divergence` per base (a star phylogeny: transposon families, segmental duplications, the
                 variant bubbles of a pangenome). An m-mer of the consensus survives in copies * (1 - divergence)^m
                 copies, its one-substitution variants in fewer, and so on: ONE large family yields a whole spectrum of
                 m-mer multiplicities, and a mixture of family sizes yields the heavy-tailed bucket-size distribution
                 the reference's build logs show (benchmarks/results-10-11-25/k31/regular-build.log);
  * cores        the limit of large divergence: `copies` unrelated random strings sharing `core` consecutive bases
                 (m <= core < k): short exact repeats in unrelated contexts (low-complexity sequence, the conserved heart
                 of an ancient repeat). Every m-mer inside the core occurs `copies` times in distinct k-mers; which of them
                 become large buckets depends on the minimizer order in use -- any order sees the same distribution;
  * de-duplication  a spectrum-preserving string set holds every k-mer ONCE (either strand), so the k-mers of the family
                 copies are sorted (canonical form), every occurrence after the first is dropped, and each copy falls
                 apart into the maximal runs of k-mers it still owns -- one string per run, as the branches of a
                 de Bruijn graph fall apart into unitigs. What remains around a substitution is a short string whose
                 m-mers are shared with the other copies while its k-mers are its own: a bucket of several positions.

The amounts of each family class are fitted (tools/calibrate_repeats.py: non-negative least squares over per-class
bucket histograms measured by building each class alone) to the statistics the reference printed for the real
collection: RECIPES below. Everything runs on torch tensors (CPU here, the GPU on the bench box).
"""
from __future__ import annotations

import json
import os

import numpy as np

_M64 = (1 << 64) - 1


def _s64(v: int) -> int:
    return v - (1 << 64) if v >= (1 << 63) else v


def _shr(x, s: int):
    return (x >> s) & ((1 << (64 - s)) - 1) if s else x


def _reverse_pairs64(x):
    x = ((x & 0x3333333333333333) << 2) | (_shr(x, 2) & 0x3333333333333333)
    x = ((x & 0x0F0F0F0F0F0F0F0F) << 4) | (_shr(x, 4) & 0x0F0F0F0F0F0F0F0F)
    x = ((x & 0x00FF00FF00FF00FF) << 8) | (_shr(x, 8) & 0x00FF00FF00FF00FF)
    x = ((x & 0x0000FFFF0000FFFF) << 16) | (_shr(x, 16) & 0x0000FFFF0000FFFF)
    return (x << 32) | _shr(x, 32)


def _canonical_kmers(codes, k: int):
    """codes: (R, L) uint8 tensor -> the canonical k-mers of every row, 2-bit packed with the first base in the low bits
    (reference include/kmer.hpp:80,159-165): k <= 31: one (R, L-k+1) int64 tensor, min(x, revcomp(x)); k <= 63: the pair
    (high word, low word) of whichever of x and revcomp(x) comes first in (high, low) order -- any total order does, the
    pair only has to be the same for a k-mer and its reverse complement."""
    import torch

    R, L = codes.shape
    n = L - k + 1
    # (zero columns behind the row: the 16-base words of the last k-mers run past the row's end)
    c = torch.cat([codes, torch.zeros((R, 64), dtype=codes.dtype, device=codes.device)], dim=1).to(torch.int64)
    h = torch.zeros((R, L + 48), dtype=torch.int64, device=codes.device)  # h[j] = the 16 bases starting at j
    for t in range(16):
        h |= c[:, t:t + L + 48] << (2 * t)
    comp = _s64(0xAAAAAAAAAAAAAAAA)

    def word(at, bases):  # `bases` (<= 32) bases starting `at` bases into every k-mer, as one int64
        lo = h[:, at:at + n]
        if bases <= 16:
            return lo & ((1 << (2 * bases)) - 1)
        hi = h[:, at + 16:at + 16 + n]
        if bases < 32:
            hi = hi & ((1 << (2 * (bases - 16))) - 1)
        return lo | (hi << 32)

    if k <= 31:
        x = word(0, k)
        rc = _shr(_reverse_pairs64(x ^ comp), 64 - 2 * k)
        return torch.minimum(x, rc)
    lo, hi = word(0, 32), word(32, k - 32)
    r_hi, r_lo = _reverse_pairs64(lo ^ comp), _reverse_pairs64(hi ^ comp)  # the words swap (kmer.hpp:162)
    s = 128 - 2 * k  # 2 <= s <= 62
    rc_lo = _shr(r_lo, s) | (r_hi << (64 - s))
    rc_hi = _shr(r_hi, s)
    first = (rc_hi < hi) | ((rc_hi == hi) & (rc_lo < lo))
    return torch.where(first, rc_hi, hi), torch.where(first, rc_lo, lo)


def _first_occurrences(canon):
    """-> bool tensor over the flattened k-mer starts: True where a canonical k-mer occurs for the first time (lowest row, then
    lowest position: stable sorts)."""
    import torch

    if not isinstance(canon, tuple):
        flat = canon.reshape(-1)
        order = torch.sort(flat, stable=True)
        first = torch.ones_like(order.values, dtype=torch.bool)
        first[1:] = order.values[1:] != order.values[:-1]
        keep = torch.zeros(flat.numel(), dtype=torch.bool, device=flat.device)
        keep[order.indices] = first
        return keep
    hi, lo = canon[0].reshape(-1), canon[1].reshape(-1)
    by_lo = torch.sort(lo, stable=True).indices
    by_hi = torch.sort(hi[by_lo], stable=True).indices
    perm = by_lo[by_hi]  # (high, low) order, ties in input order
    h, l = hi[perm], lo[perm]
    first = torch.ones(perm.numel(), dtype=torch.bool, device=perm.device)
    first[1:] = (h[1:] != h[:-1]) | (l[1:] != l[:-1])
    keep = torch.zeros(perm.numel(), dtype=torch.bool, device=perm.device)
    keep[perm] = first
    return keep


def _family_strings(gen, device, families: int, copies: int, length: int, divergence: float, k: int, core: int = 0,
                    max_kmers: int = 1 << 25, salt: int = 0):
    """-> (codes uint8 1-D, lengths int64 1-D): the strings left of `families` families after de-duplication.
    core > 0: the copies of a family are unrelated random strings sharing only `core` consecutive bases (in the middle)
    instead of diverged copies of one consensus."""
    import torch

    out_codes, out_lens = [], []
    per = max(1, max_kmers // max(1, copies * (length - k + 1)))  # families per chunk (duplicates never cross families)
    done = 0
    while done < families:
        F = min(per, families - done)
        done += F
        if core > 0:
            # drawn on the CPU whatever the device: which cores become the largest buckets is a matter of their hash, the few
            # large core families were chosen seed by seed for that (calibrate_repeats.py TAIL_TUNING), and a CUDA generator
            # would draw other cores from the same seed
            cpu_gen = torch.Generator()
            # (`salt` tells the calls of one class apart: the whole families and the fractional one must not draw the same
            # flanks -- they did at --repeat-scale 0.5, and bench.py's parity check found the duplicated k-mers)
            cpu_gen.manual_seed((gen.initial_seed() * 4 + salt) * 1009 + 31 * done)
            rows = torch.randint(0, 4, (F, copies, length), generator=cpu_gen, dtype=torch.uint8)
            at = (length - core) // 2
            rows[:, :, at:at + core] = torch.randint(0, 4, (F, 1, core), generator=cpu_gen, dtype=torch.uint8)
            rows = rows.reshape(F * copies, length).to(device)
        else:
            cons = torch.randint(0, 4, (F, 1, length), generator=gen, device=device, dtype=torch.uint8)
            rows = cons.expand(F, copies, length).reshape(F * copies, length).clone()
            mut = torch.rand(rows.shape, generator=gen, device=device) < divergence
            delta = torch.randint(1, 4, rows.shape, generator=gen, device=device, dtype=torch.uint8)
            rows = torch.where(mut, (rows + delta) & 3, rows)
            del mut, delta
        R = rows.shape[0]
        n = length - k + 1
        keep = _first_occurrences(_canonical_kmers(rows, k)).reshape(R, n)  # every canonical k-mer once
        # maximal runs of kept k-mer starts inside a row
        pad = torch.zeros((R, 1), dtype=torch.bool, device=device)
        a = torch.cat([pad, keep, pad], dim=1).to(torch.int8)
        d = a[:, 1:] - a[:, :-1]  # (R, n+1): +1 at a run's first start, -1 one past its last
        starts = (d == 1).nonzero()  # (runs, 2) row-major order: runs of a row are in order
        ends = (d == -1).nonzero()
        count = ends[:, 1] - starts[:, 1]  # k-mers in the run
        lens = count + (k - 1)
        first_base = starts[:, 0] * length + starts[:, 1]
        total = int(lens.sum().item())
        excl = torch.cumsum(lens, 0) - lens
        idx = torch.repeat_interleave(first_base - excl, lens) + torch.arange(total, device=device)
        out_codes.append(rows.reshape(-1)[idx])
        out_lens.append(lens)
    import torch as _t

    return _t.cat(out_codes), _t.cat(out_lens)


def _duplicate_kmer_starts(codes, lens, k: int, device, chunk: int = 1 << 26, parts: int = 16) -> np.ndarray:
    """codes: all strings back to back (uint8, CPU), lens: their lengths; k <= 31 -> the start positions (ascending) of every k-mer
    whose canonical form occurred at a lower position already. The k-mers are made on `device` chunk by chunk and compared part by
    part (a hash of the k-mer picks the part: a sort of 2.5 x 10^9 values in one piece is more than a sort takes)."""
    import torch

    N = int(codes.numel())
    n = N - k + 1
    if n <= 0:
        return np.zeros(0, dtype=np.int64)
    vals = torch.empty(n, dtype=torch.int64, device=device)
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        vals[a:b] = _canonical_kmers(codes[a:b + k - 1].to(device).reshape(1, -1), k).reshape(-1)
    valid = torch.ones(n, dtype=torch.bool, device=device)  # a start within k - 1 bases of its string's end is no k-mer
    ends = torch.cumsum(lens.to(device), 0)
    for a in range(0, int(ends.numel()), 1 << 22):
        idx = (ends[a:a + (1 << 22), None] - torch.arange(1, k, device=device)[None, :]).reshape(-1)
        valid[idx[(idx >= 0) & (idx < n)]] = False
    del ends
    part = ((vals * _s64(0x9E3779B97F4A7C15)) >> 40) & (parts - 1)  # (bits of the product's upper half: the low bits of a k-mer are its first bases)
    found = []
    for j in range(parts):
        # (torch.nonzero takes fewer than 2^31 elements at a time)
        pos = torch.cat([torch.nonzero((part[a:a + (1 << 30)] == j) & valid[a:a + (1 << 30)])[:, 0] + a for a in range(0, n, 1 << 30)])
        if pos.numel() < 2:
            continue
        order = torch.sort(vals[pos], stable=True)  # equal k-mers in order of position
        again = order.values[1:] == order.values[:-1]
        if bool(again.any()):
            found.append(pos[order.indices[1:][again]].cpu().numpy())
    return np.sort(np.concatenate(found)) if found else np.zeros(0, dtype=np.int64)


def _cut_kmers_out(codes, lens, starts: np.ndarray, k: int):
    """The strings without the k-mers that start at `starts` (a handful): the string holding one falls apart into what lies before
    the k-mer's last base and what lies behind its first -- the two pieces keep every other k-mer --, a piece shorter than k is dropped."""
    import torch

    lens = lens.numpy().copy()
    for p in sorted((int(v) for v in starts), reverse=True):  # from the back: what lies before stays where it is
        ends = np.cumsum(lens)
        s = int(np.searchsorted(ends, p, side="right"))
        begin, end = (int(ends[s - 1]) if s else 0), int(ends[s])
        left, right = codes[begin:p + k - 1], codes[p + 1:end]
        pieces = [piece for piece in (left, right) if piece.numel() >= k]
        codes = torch.cat([codes[:begin]] + pieces + [codes[end:]])
        lens = np.concatenate([lens[:s], np.array([int(piece.numel()) for piece in pieces], dtype=lens.dtype), lens[s + 1:]])
    return codes, torch.from_numpy(lens)


def pack_codes_torch(codes):
    """uint8 base codes (0..3) on any device -> numpy uint64 words, base i in bits [2i, 2i+1] of word i // 32."""
    import torch

    n = codes.numel()
    pad = (-n) % 32
    if pad:
        codes = torch.cat([codes, torch.zeros(pad, dtype=torch.uint8, device=codes.device)])
    q = codes.reshape(-1, 4)
    b = q[:, 0] | (q[:, 1] << 2) | (q[:, 2] << 4) | (q[:, 3] << 6)
    return np.ascontiguousarray(b.cpu().numpy()).view("<u8").copy()


def _background(gen, rng, dev, bases: int, mean_len: float, k: int):
    import torch

    est = int(bases / mean_len * 1.05) + 16
    lens = k + rng.geometric(1.0 / max(1.0, mean_len - k + 1), est) - 1
    csum = np.cumsum(lens)
    n_str = int(np.searchsorted(csum, bases)) + 1
    lens = torch.from_numpy(lens[:n_str].astype(np.int64)).to(dev)
    total = int(lens.sum().item())
    chunks = []
    for at in range(0, total, 1 << 28):
        chunks.append(torch.randint(0, 4, (min(1 << 28, total - at),), generator=gen, device=dev, dtype=torch.uint8))
    return torch.cat(chunks), lens


def make_repeat_spss(num_bases: int, k: int = 31, classes=(), seed: int = 0x5555AAAA, mean_len: float = 400.0,
                     reference_bases: float = 1e9, background=None, device=None):
    """-> (packed words, endpoints) of about `num_bases` bases.

    classes: iterable of dicts {"copies", "length", "divergence" | "core", "families"}; "families" is the number of families per
    `reference_bases` bases of output, scaled to num_bases; a fractional amount becomes one more family with that fraction
    of the copies. background: list of {"mean_len", "bases"} (bases per reference_bases) -- random
    strings of length k + geometric; None: one background of mean length `mean_len` fills what the families left of
    num_bases. Deterministic for (arguments, device type)."""
    import torch

    if k > 63:
        raise ValueError("make_repeat_spss: k <= 63")
    dev = torch.device(device) if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(seed) & 0x7FFFFFFFFFFFFFFF)
    rng = np.random.default_rng(seed)
    scale = num_bases / reference_bases
    parts_codes, parts_lens = [], []
    used = 0
    for ci, c in enumerate(classes):
        # every class draws from its own generator (seed, the class's "seed" field or a number made of its parameters): the realisation of a
        # class does not depend on the classes before it, so the few largest families -- whose luck with the minimizer order
        # decides the largest buckets -- can be examined, and chosen, on their own (tools/calibrate_repeats.py --tune-tail)
        own = c.get("seed")
        if own is None:  # from the class's parameters, not from its place in the list: a recipe refitted with one class more or
            # less keeps the realisations of the others
            own = (int(c["copies"]) * 131 + int(c["length"]) * 7 + int(round(float(c.get("divergence", 0.0)) * 1000)) * 3 + int(c.get("core", 0)) * 17) % 100003
        gen.manual_seed((int(seed) * 1000003 + int(own) * 7919 + 17) & 0x7FFFFFFFFFFFFFFF)
        want = float(c["families"]) * scale
        copies = int(c["copies"])
        # floor(want) whole families, and the remaining fraction as ONE family with that fraction of the copies (a class
        # of a few huge families cannot be rounded to whole families without moving the totals by tens of per cent)
        todo = [(int(want), copies)]
        part = int(round((want - int(want)) * copies))
        if part >= 2:
            todo.append((1, part))
        for call, (F, n) in enumerate(todo):
            if F == 0:
                continue
            codes, lens = _family_strings(gen, dev, F, n, int(c["length"]), float(c.get("divergence", 0.0)), k,
                                          core=int(c.get("core", 0)), salt=call)
            parts_codes.append(codes.cpu())
            parts_lens.append(lens.cpu())
            used += int(codes.numel())
    if background is None:
        background = [{"mean_len": mean_len, "bases": max(0, num_bases - used) / scale}] if num_bases - used >= k else []
    gen.manual_seed((int(seed) * 1000003 + 999983) & 0x7FFFFFFFFFFFFFFF)
    for b in background:
        want = int(float(b["bases"]) * scale)
        if want < k:
            continue
        codes, lens = _background(gen, rng, dev, want, float(b["mean_len"]), k)
        parts_codes.append(codes.cpu())
        parts_lens.append(lens.cpu())
    codes = torch.cat(parts_codes)
    lens = torch.cat(parts_lens)
    if k <= 31:
        # the de-duplication above works family by family; two k-mers of DIFFERENT families (or of the background) coincide by chance
        # about n^2 / 4^k times -- once or twice among the 0.9 / 2.5 x 10^9 k-mers of the C2 / C3 stand-ins (round 5's full-size
        # lookup(access(id)) == id test met the one of C2). A spectrum-preserving string set holds every k-mer once: the later
        # occurrences are cut out of their strings. (k = 63: 4^63 leaves no such chance.)
        extra = _duplicate_kmer_starts(codes, lens, k, dev)
        if extra.size:
            codes, lens = _cut_kmers_out(codes, lens, extra, k)
    endpoints = np.zeros(int(lens.numel()) + 1, dtype=np.uint64)
    endpoints[1:] = np.cumsum(lens.numpy()).astype(np.uint64)
    return pack_codes_torch(codes), endpoints


def make_recipe_spss(name: str, num_bases: int | None = None, seed: int = 0x5555AAAA, device=None):
    """The stand-in of a named collection at `num_bases` bases (default: the collection's own size)."""
    r = load_recipe(name)
    n = int(num_bases if num_bases is not None else r["reference_bases"])
    return make_repeat_spss(n, k=int(r["k"]), classes=r["classes"], seed=seed, reference_bases=float(r["reference_bases"]),
                            background=r["background"], device=device)


def load_recipe(name: str) -> dict:
    """The fitted class amounts for a named collection (sshash_amd/recipes/<name>.json, written by
    tools/calibrate_repeats.py together with the published targets they were fitted to)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "recipes", name + ".json")
    with open(path) as f:
        return json.load(f)


def statistics_vs_target(stats: dict, name: str) -> dict:
    """Bucket statistics of a built stand-in (Dictionary.bucket_stats()) next to the numbers the reference printed for the
    real collection (recipe["target"], scaled by the ratio of the base counts when the stand-in is smaller):
    {statistic: {"target", "achieved", "ratio"}}."""
    t = load_recipe(name)["target"]
    scale = stats["num_bases"] / t["num_bases"]
    out = {"scale": round(scale, 6), "source": t["source"]}

    def row(key, target, achieved, scaled=True):
        tt = target * scale if scaled else target
        out[key] = {"target": round(tt, 1), "achieved": achieved, "ratio": round(achieved / tt, 4) if tt else None}

    for key in ("num_kmers", "num_strings", "num_minimizers", "num_minimizer_positions",
                "num_buckets_larger_than_1_not_in_skew_index", "num_minimizer_positions_of_buckets_larger_than_1",
                "num_buckets_in_skew_index", "num_minimizer_positions_of_buckets_in_skew_index", "num_kmers_in_skew_index"):
        row(key, t[key], stats[key])
    row("max_bucket_size", t["max_bucket_size"], stats["max_bucket_size"], scaled=False)
    part = list(stats["num_kmers_in_skew_partition"]) + [0] * 8
    for p, v in enumerate(t["num_kmers_in_skew_partition"]):
        row(f"num_kmers_in_skew_partition_{p}", v, part[p])
    for s, pct in enumerate(t["bucket_percent"], start=1):
        row(f"buckets_with_{s}_positions", pct / 100.0 * t["num_minimizers"], stats["buckets_with_n_positions"][s - 1])
    return out
