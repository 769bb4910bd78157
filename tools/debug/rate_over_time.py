#!/usr/bin/env python
"""One process, ONE upload of the C3 dictionary: the lookup rate measured again and again (bursts of 8 calls over 2.5e8 queries), back to back and
after pauses -- is the 37 / 40.5 G lookups/s difference a property of the replica's memory, or of the moment?   python tools/debug/rate_over_time.py"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench, sshash_amd
from sshash_amd.repeats import load_recipe
from sshash_amd.synthetic import draw_queries_device

bases, recipe, _, _ = bench.WORKLOADS["c3"]
r = load_recipe(recipe)
args = argparse.Namespace(bases=bases, k=int(r["k"]), m=int(r["m"]), recipe=recipe, repeat_scale=1.0, canonical=False, seed=0x5555AAAA,
                          cache_dir=os.environ.get("SSHASH_BENCH_CACHE", "/tmp"), verbose=False)
d, path = bench.get_index(args, 0, 1, lambda: None)
d.to_device(0)
n = 250_000_000
dq = draw_queries_device(d, 0, n, 0.5, seed=7)
out = torch.empty(n, dtype=torch.int64, device="cuda:0")
s = torch.cuda.current_stream()
t_start = time.time()


def burst(calls=8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(calls):
        d.lookup_device(0, dq.data_ptr(), n, out.data_ptr(), stream=s.cuda_stream)
    e1.record(s)
    torch.cuda.synchronize()
    return n * calls / e0.elapsed_time(e1) / 1e6


burst(2)
rates = []
for pause in [0] * 12 + [0.5] * 6 + [3.0] * 4 + [0] * 8 + [10.0] * 2 + [0] * 6:
    if pause:
        time.sleep(pause)
    rates.append((round(time.time() - t_start, 1), pause, round(burst(), 2)))
print(f"process {os.getpid()}:", " ".join(f"{t}s{'(after %.1fs idle)' % p if p else ''}:{r}" for t, p, r in rates), flush=True)
