#!/usr/bin/env python
"""One process, the same dictionary uploaded again and again: does the lookup rate of a replica vary from upload to upload (then it is where
the allocation lands, and an upload could try again), or only from process to process?   python tools/debug/upload_modes.py [uploads]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench, sshash_amd
from sshash_amd.repeats import load_recipe
from sshash_amd.synthetic import draw_queries_device

uploads = int(sys.argv[1]) if len(sys.argv) > 1 else 5
bases, recipe, _, _ = bench.WORKLOADS["c3"]
r = load_recipe(recipe)
args = argparse.Namespace(bases=bases, k=int(r["k"]), m=int(r["m"]), recipe=recipe, repeat_scale=1.0, canonical=False, seed=0x5555AAAA,
                          cache_dir=os.environ.get("SSHASH_BENCH_CACHE", "/tmp"), verbose=False)
d, path = bench.get_index(args, 0, 1, lambda: None)
n = 250_000_000
out = None
for i in range(uploads):
    if i:
        d = sshash_amd.Dictionary.load(path)
    t0 = time.time()
    d.to_device(0)
    up = time.time() - t0
    if out is None:
        dq = draw_queries_device(d, 0, n, 0.5, seed=7)
        out = torch.empty(n, dtype=torch.int64, device="cuda:0")
    s = torch.cuda.current_stream()
    for _ in range(2):
        d.lookup_device(0, dq.data_ptr(), n, out.data_ptr(), stream=s.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(8):
        d.lookup_device(0, dq.data_ptr(), n, out.data_ptr(), stream=s.cuda_stream)
    e1.record(s)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 8
    print(f"process {os.getpid()} upload {i}: {n / ms / 1e6:.2f} G lookups/s ({ms:.3f} ms per 2.5e8 queries; upload {up:.1f} s)", flush=True)
    d.close()
