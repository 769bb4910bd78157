#!/usr/bin/env python
"""Debug aid: false negatives of the table path on a synthetic human-like SPSS with MIXED batches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import sshash_amd
from sshash_amd.synthetic import make_spss, revcomp_device
from oracle.ground_truth import _revcomp_u64

bases = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000_000
k, m = 31, 21
words, endpoints = make_spss(bases, k=k, m=m, mean_len=274.0)
d = sshash_amd.Dictionary.build_from_packed(words, endpoints, k=k, m=m, num_threads=0)
d.to_device(0)
print(d.device_stats(0))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(5)
ids = torch.randint(0, d.num_kmers(), (n,), generator=g, device=dev, dtype=torch.int64)
pos = torch.empty((n, 1), dtype=torch.int64, device=dev)
d.access_packed_device(0, ids.data_ptr(), n, pos.data_ptr())
flip = torch.rand(n, generator=g, device=dev) < 0.5
pos = torch.where(flip.unsqueeze(1), revcomp_device(pos, k), pos)
for frac in (1.0, 0.5):
    is_pos = torch.rand(n, generator=g, device=dev) < frac
    neg = (torch.randint(0, 1 << 31, (n,), generator=g, device=dev, dtype=torch.int64) << 31) | torch.randint(0, 1 << 31, (n,), generator=g, device=dev, dtype=torch.int64)
    q = torch.where(is_pos, pos[:, 0], neg).contiguous()
    out = torch.empty(n, dtype=torch.int64, device=dev)
    for rep in range(2):
        d.lookup_device(0, q.data_ptr(), n, out.data_ptr())
        torch.cuda.synchronize()
        bad = is_pos & (out != ids)
        print("positive fraction", frac, "rep", rep, "false negatives / wrong ids", int(bad.sum().item()), "of", int(is_pos.sum().item()))
    if int(bad.sum().item()):
        bi = torch.nonzero(bad)[:2000, 0].cpu().numpy()
        bq = q[bi].cpu().numpy().view(np.uint64)
        # table key of the failing k-mers
        def mm_hash(x):
            return (np.uint32(x & 0xFFFFFFFF) * np.uint32(0x9E3779B1) + (np.uint32(x >> 32) * np.uint32(0x85EBCA77) + np.uint32(0x27D4EB2F))) & np.uint32(0xFFFFFFFF)
        x = bq; xr = _revcomp_u64(x, k); mask = np.uint64((1 << (2 * m)) - 1)
        bf = np.full(x.size, 0xFFFFFFFF, dtype=np.uint64); br = bf.copy(); pf = np.zeros(x.size, dtype=np.int64); pr = pf.copy()
        f, r = x.copy(), xr.copy()
        with np.errstate(over="ignore"):
            for i in range(k - m + 1):
                hf = mm_hash(f & mask).astype(np.uint64); hr = mm_hash(r & mask).astype(np.uint64)
                u = hf < bf; bf[u] = hf[u]; pf[u] = i
                u = hr < br; br[u] = hr[u]; pr[u] = i
                f >>= np.uint64(2); r >>= np.uint64(2)
        rc = br < bf
        key = (np.where(rc, xr, x) >> (2 * np.where(rc, pr, pf)).astype(np.uint64)) & mask
        keyrc = _revcomp_u64(key, m)
        # occurrences of the keys as m-mers of the strings (both strands)
        nb = int(endpoints[-1])
        codes = np.zeros(nb, dtype=np.uint8)
        wv = words
        for i in range(32):
            sel = np.arange(i, nb, 32)
            codes[sel] = ((wv[: sel.size] >> np.uint64(2 * i)) & np.uint64(3)).astype(np.uint8)
        mm = np.zeros(nb - m + 1, dtype=np.uint64)
        for i in range(m):
            mm |= codes[i: nb - m + 1 + i].astype(np.uint64) << np.uint64(2 * i)
        uk = np.unique(np.concatenate([key, keyrc]))
        hits = mm[np.isin(mm, uk)]
        vals, cnts = np.unique(hits, return_counts=True)
        cnt = dict(zip(vals.tolist(), cnts.tolist()))
        occ = np.array([cnt.get(int(a), 0) + (cnt.get(int(b), 0) if a != b else 0) for a, b in zip(key, keyrc)])
        print("failing k-mers by m-mer occurrences of their key: 1:", int((occ == 1).sum()), "2-4:", int(((occ > 1) & (occ <= 4)).sum()), "5-64:", int(((occ > 4) & (occ <= 64)).sum()), ">64:", int((occ > 64).sum()), "max", int(occ.max()))
        print("got values sample", out[torch.from_numpy(bi[:8]).to(dev)].cpu().numpy(), "want", ids[torch.from_numpy(bi[:8]).to(dev)].cpu().numpy())
