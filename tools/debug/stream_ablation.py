#!/usr/bin/env python
"""Time the streaming kernels of whatever library SSHASH_AMD_LIBRARY names on the bench's high-hit / C4 read sets, WITHOUT checking the
counters (the ablation variants -- no walk past a first bucket, no run measurement, no skipping -- count wrongly on purpose: what is
asked is what each part costs)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from sshash_amd.repeats import load_recipe
from sshash_amd.synthetic import make_reads_device

workload, positive = sys.argv[1], float(sys.argv[2])
bases, recipe, _, _ = bench.WORKLOADS[workload]
r = load_recipe(recipe)
args = argparse.Namespace(bases=bases, k=int(r["k"]), m=int(r["m"]), recipe=recipe, repeat_scale=1.0, canonical=False, seed=0x5555AAAA, cache_dir="/tmp", verbose=False)
d, _ = bench.get_index(args, 0, 1, lambda: None)
d.to_device(0)
n, L = 20_000_000, 150
reads = make_reads_device(d, 0, n, L, positive_fraction=positive, seed=args.seed)
offsets = torch.arange(n + 1, dtype=torch.int64, device="cuda:0") * L
report = torch.zeros(6, dtype=torch.int64, device="cuda:0")
for _ in range(2):
    d.streaming_query_device(0, reads.data_ptr(), offsets.data_ptr(), n, report.data_ptr())
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
report.zero_()
e0.record()
for _ in range(5):
    d.streaming_query_device(0, reads.data_ptr(), offsets.data_ptr(), n, report.data_ptr())
e1.record()
torch.cuda.synchronize()
print(os.path.basename(os.environ.get("SSHASH_AMD_LIBRARY", "shipped")), workload, positive, round(e0.elapsed_time(e1) / 5, 3), "ms", [int(v) // 5 for v in report.cpu().tolist()], flush=True)
