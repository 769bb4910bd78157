#!/usr/bin/env python
"""Which ids of the full-size C2 dictionary fail lookup(access(id)) == id, and why: is the k-mer spelled twice by the stand-in's strings
(then both ids are right and the SPSS is not one), or does the lookup miss / err?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from test_gpu_baseline_workloads import full_size_dictionary
from oracle import oracle as O
from sshash_amd.synthetic import revcomp_device

workload = sys.argv[1] if len(sys.argv) > 1 else "c2"
d, path, args = full_size_dictionary(workload)
k, W = args.k, 1 if args.k <= 31 else 2
dev = torch.device("cuda", 0)
n = d.num_kmers()
step = 200_000_000
bad_all = []
for lo in range(0, n, step):
    m = min(step, n - lo)
    ids = torch.arange(lo, lo + m, dtype=torch.int64, device=dev)
    q = torch.empty((m, W), dtype=torch.int64, device=dev)
    d.access_packed_device(0, ids.data_ptr(), m, q.data_ptr())
    out = torch.empty(m, dtype=torch.int64, device=dev)
    for strand, qq in (("fwd", q), ("rc", revcomp_device(q, k).contiguous())):
        d.lookup_device(0, qq.data_ptr(), m, out.data_ptr())
        torch.cuda.synchronize()
        bad = torch.nonzero(out != ids)[:, 0]
        if bad.numel():
            got = out[bad].contiguous()
            back = torch.full((bad.numel(), W), -1, dtype=torch.int64, device=dev)
            found = got != -1
            if bool(found.any()):
                gi = got[found].contiguous()
                bk = torch.empty((gi.numel(), W), dtype=torch.int64, device=dev)
                d.access_packed_device(0, gi.data_ptr(), gi.numel(), bk.data_ptr())
                back[found] = bk
            torch.cuda.synchronize()
            for j in range(min(20, bad.numel())):
                i = int(bad[j])
                asked = qq[i].cpu().numpy().view(np.uint64)
                bad_all.append((strand, lo + i, int(got[j]), [hex(int(v)) for v in asked], [hex(int(v) & (2**64 - 1)) for v in back[j].cpu().numpy()]))
            print(strand, "ids", lo, "..", lo + m, ":", int(bad.numel()), "differ", flush=True)
print("total differing (first 20 per chunk listed):", len(bad_all))
ora = O.OracleIndex(path)
for strand, want, got, asked, back in bad_all:
    kmer = np.array([int(a, 16) for a in asked], dtype=np.uint64)
    oid = ora.lookup_ids(kmer, num_threads=1)
    print(strand, "id", want, "GPU returned", got, "oracle returns", int(oid[0]) if int(oid[0]) != 2**64 - 1 else "absent", "asked", asked, "access(GPU's id)", back)
