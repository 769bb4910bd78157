#!/usr/bin/env python
"""Debug aid: how fast does sshash_streaming_query take reads that already sit in host memory (no file, no parsing)?
The consumer side of sshash_streaming_query_from_file: staging into pinned lanes, H2D, streaming kernels."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import argparse as A
import numpy as np
import bench
from sshash_amd import _binding as B
from sshash_amd.synthetic import make_reads_device

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
b = A.Namespace(bases=2_813_192_630, k=31, m=21, canonical=False, seed=0x5555AAAA, cache_dir="/tmp", verbose=False, recipe="human_k31", repeat_scale=1.0)
d, _ = bench.get_index(b, 0, 1, lambda: None)
d.to_device(0)
reads = make_reads_device(d, 0, n, 150, positive_fraction=0.9).cpu().numpy()
bases = np.ascontiguousarray(reads).reshape(-1)
offsets = (np.arange(n + 1, dtype=np.uint64) * np.uint64(150))
rep = B._Report()
for turn in range(3):
    t0 = time.perf_counter()
    B._check(B._load().sshash_streaming_query(d._h, bases.ctypes.data, offsets.ctypes.data, n, C.byref(rep)))
    dt = time.perf_counter() - t0
    print(f"turn {turn}: {dt:.3f} s, {bases.size / dt / 1e9:.2f} GB/s of bases, {rep.num_kmers / dt / 1e9:.2f} G k-mers/s", flush=True)
