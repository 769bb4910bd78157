#!/usr/bin/env python
"""Debug aid: which k-mers does the table path get wrong? Looks up every k-mer of a FASTA (both strands) on cuda:0 and
classifies the failures by the table's own key function (number of occurrences of the k-mer's key)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import sshash_amd
from oracle.ground_truth import GroundTruth, _revcomp_u64, read_fasta_sequences

fasta = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests/golden/salmonella_enterica_k31_ust.fa.gz")
k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
m = int(sys.argv[3]) if len(sys.argv) > 3 else 13
d = sshash_amd.Dictionary.build(fasta, k=k, m=m, num_threads=0)
d.to_device(0)
print(d.device_stats(0))
gt = GroundTruth(read_fasta_sequences(fasta, k), k)
n = gt.num_kmers
ids = np.arange(n, dtype=np.uint64)
q = gt.kmers(ids)
for name, qq in (("fwd", q), ("rc", _revcomp_u64(q, k))):
    dq = torch.from_numpy(qq.view(np.int64)).cuda()
    out = torch.empty(n, dtype=torch.int64, device="cuda")
    d.lookup_device(0, dq.data_ptr(), n, out.data_ptr())
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint64)
    bad = np.nonzero(got != ids)[0]
    print(name, "mismatches", bad.size, "of", n, "first", bad[:10], "got", got[bad[:10]])

# the table's key function (device_layout.hpp sk_key), k <= 31
def mm_hash(x):
    return (np.uint32(x & 0xFFFFFFFF) * np.uint32(0x9E3779B1) + (np.uint32(x >> 32) * np.uint32(0x85EBCA77) + np.uint32(0x27D4EB2F))) & np.uint32(0xFFFFFFFF)
def keys_of(x):
    xr = _revcomp_u64(x, k)
    mask = np.uint64((1 << (2 * m)) - 1)
    bf = np.full(x.size, 0xFFFFFFFF, dtype=np.uint64); br = bf.copy()
    pf = np.zeros(x.size, dtype=np.int64); pr = pf.copy()
    f, r = x.copy(), xr.copy()
    with np.errstate(over="ignore"):
        for i in range(k - m + 1):
            hf = mm_hash(f & mask).astype(np.uint64); hr = mm_hash(r & mask).astype(np.uint64)
            u = hf < bf; bf[u] = hf[u]; pf[u] = i
            u = hr < br; br[u] = hr[u]; pr[u] = i
            f >>= np.uint64(2); r >>= np.uint64(2)
    rc = br < bf; tie = br == bf
    pos = np.where(rc, pr, pf)
    src = np.where(rc, xr, x)
    key = (src >> (2 * pos).astype(np.uint64)) & mask
    return key, pos, rc, tie
key, pos, rc, tie = keys_of(q)
# occurrence = (string position of the key); k-mer i starts at base offset gt.offsets? use id -> offset via strings
off = gt.endpoints[gt.string_id.astype(np.int64)] + gt.in_string.astype(np.uint64)
occ = off.astype(np.int64) + np.where(rc, (k - m) - pos, pos)
pairs = np.unique(np.stack([key, occ.astype(np.uint64)], 1), axis=0)
cnt = collections.Counter(pairs[:, 0].tolist())
occs = np.array([cnt[int(x)] for x in key])
print("k-mers by occurrences of their key: 1:", int((occs == 1).sum()), "2-4:", int(((occs > 1) & (occs <= 4)).sum()), ">4:", int((occs > 4).sum()), "ties:", int(tie.sum()))
for name, qq in (("fwd", q),):
    dq = torch.from_numpy(qq.view(np.int64)).cuda(); out = torch.empty(n, dtype=torch.int64, device="cuda")
    d.lookup_device(0, dq.data_ptr(), n, out.data_ptr()); torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint64); bad = got != ids
    print("bad by class: 1:", int((bad & (occs == 1)).sum()), "2-4:", int((bad & (occs > 1) & (occs <= 4)).sum()), ">4:", int((bad & (occs > 4)).sum()), "tie:", int((bad & tie).sum()))
    b = np.nonzero(bad)[0][:5]
    for i in b:
        print("id", i, "got", got[i], "occs of key", occs[i], "pos", pos[i], "rc", rc[i], "tie", tie[i], "neighbours bad", bad[max(0, i - 3): i + 4])


dq = torch.from_numpy(q.view(np.int64)).cuda()
for rep in range(3):
    mo = torch.full((n,), 0xAB, dtype=torch.uint8, device="cuda")
    d.is_member_device(0, dq.data_ptr(), n, mo.data_ptr())
    torch.cuda.synchronize()
    got = mo.cpu().numpy()
    bad = got != 1
    vals, cnts = np.unique(got[bad], return_counts=True)
    print("device member rep", rep, "bad", int(bad.sum()), "values at bad positions", dict(zip(vals.tolist(), cnts.tolist())),
          "by class 1:", int((bad & (occs == 1)).sum()), "2-4:", int((bad & (occs > 1) & (occs <= 4)).sum()), ">4:", int((bad & (occs > 4)).sum()), "tie:", int((bad & tie).sum()))
    b = np.nonzero(bad)[0]
    print("   first bad", b[:12], "gaps", np.diff(b[:12]))
