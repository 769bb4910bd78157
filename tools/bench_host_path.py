#!/usr/bin/env python
"""Host-buffer entry point (sshash_lookup_packed: pageable caller arrays, PCIe inclusive) on the bench
dictionary, output array allocated and touched once -- what a C/C++ caller that reuses its buffers sees.
SSHASH_AMD_TEST_HOOKS="host_lanes=N,host_chunk=M" overrides the lane count and chunk size of the pipeline."""
import sys, os, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, sshash_amd
from sshash_amd.synthetic import draw_queries
ns = argparse.Namespace(bases=1_387_536_274, k=31, m=21, mean_len=85.0, canonical=False, seed=0x5555AAAA, cache_dir='/tmp', verbose=False)
d, path = bench.get_index(ns, 0, 1, lambda: None)
d.to_device(0)
n = 50_000_000
q = draw_queries(d, n, 0.5, seed=5)
out = np.zeros(n, dtype=np.uint64)   # touched once
from sshash_amd import _binding as B
r = B._Results(); r.kmer_id = out.ctypes.data
lib = B._load()
def call():
    st = lib.sshash_lookup_packed(d._h, q.ctypes.data, n, 1, B.C.byref(r)); assert st == 0
call()
best = 1e9
for _ in range(4):
    t0 = time.perf_counter(); call(); best = min(best, time.perf_counter() - t0)
print(os.environ.get("SSHASH_AMD_TEST_HOOKS"), "ms", round(best*1e3, 2), "G/s", round(n/best/1e9, 3), flush=True)
