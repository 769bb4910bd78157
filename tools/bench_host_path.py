#!/usr/bin/env python
"""Host-buffer entry point (sshash_lookup_packed: pageable caller arrays, PCIe inclusive) on the bench
dictionary, output array allocated and touched once -- what a C/C++ caller that reuses its buffers sees.
SSHASH_AMD_TEST_HOOKS="host_lanes=N,host_chunk=M" overrides the lane count and chunk size of the pipeline."""
import sys, os, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, sshash_amd
from sshash_amd.synthetic import draw_queries
ns = argparse.Namespace(bases=1_387_536_274, k=31, m=21, recipe='se_k31', repeat_scale=1.0, canonical=False, seed=0x5555AAAA, cache_dir='/tmp', verbose=False)
d, path = bench.get_index(ns, 0, 1, lambda: None)
d.to_device(0)
n = 50_000_000
q = draw_queries(d, n, 0.5, seed=5)
out = np.zeros(n, dtype=np.uint64)   # touched once
from sshash_amd import _binding as B
lib = B._load()


def measure(q_ptr, out_ptr, what):
    r = B._Results(); r.kmer_id = out_ptr

    def call():
        st = lib.sshash_lookup_packed(d._h, q_ptr, n, 1, B.C.byref(r)); assert st == 0
    call()
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter(); call(); best = min(best, time.perf_counter() - t0)
    print(what, os.environ.get("SSHASH_AMD_TEST_HOOKS"), "ms", round(best * 1e3, 2), "G lookups/s", round(n / best / 1e9, 3), "GB/s over the link (16 B per lookup)", round(16 * n / best / 1e9, 1), flush=True)


measure(q.ctypes.data, out.ctypes.data, "pageable caller arrays:")
qp = torch.from_numpy(q.view(np.int64)).pin_memory()
op = torch.zeros(n, dtype=torch.int64).pin_memory()
measure(qp.data_ptr(), op.data_ptr(), "page-locked caller arrays:")
assert (op.numpy().view(np.uint64) == out).all()
# the same queries as characters (sshash_lookup_ascii: k bytes per query over the link instead of 8)
k = d.k()
chars = torch.empty((n, k), dtype=torch.uint8).pin_memory()
for a in range(0, n, 1 << 21):  # (in pieces: n x k codes as 64-bit words would be 12 GB)
    codes = (q[a:a + (1 << 21), None] >> (2 * np.arange(k, dtype=np.uint64))[None, :]) & np.uint64(3)
    chars[a:a + codes.shape[0]] = torch.from_numpy(np.frombuffer(b"ACTG", dtype=np.uint8)[codes.astype(np.int64)])
del codes
op.zero_()


def measure_ascii():
    r = B._Results(); r.kmer_id = op.data_ptr()

    def call():
        st = lib.sshash_lookup_ascii(d._h, chars.data_ptr(), n, 1, B.C.byref(r)); assert st == 0
    call()
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter(); call(); best = min(best, time.perf_counter() - t0)
    print("page-locked caller arrays, ASCII:", os.environ.get("SSHASH_AMD_TEST_HOOKS"), "ms", round(best * 1e3, 2), "G lookups/s", round(n / best / 1e9, 3),
          f"GB/s over the link ({k + 8} B per lookup)", round((k + 8) * n / best / 1e9, 1), flush=True)


measure_ascii()
assert (op.numpy().view(np.uint64) == out).all()
