#!/usr/bin/env python
"""What the lanes of the streaming kernel do per turn: a library built with -DSSHASH_STREAM_STATS (tools/ab_stats/, SSHASH_AMD_LIBRARY)
adds wave-level sums behind the six counters of the report -- turns, lanes at a fresh seed / walking / measuring a run / fetching slot 1 /
skipping invalid bases / idle / on the complete path, negatives settled without a probe. One call on the bench's read sets."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from sshash_amd.repeats import load_recipe
from sshash_amd.synthetic import make_reads_device

workload, positive = sys.argv[1], float(sys.argv[2])
bases, recipe, _, _ = bench.WORKLOADS[workload]
r = load_recipe(recipe)
args = argparse.Namespace(bases=bases, k=int(r["k"]), m=int(r["m"]), recipe=recipe, repeat_scale=1.0, canonical=False, seed=0x5555AAAA, cache_dir=os.environ.get("SSHASH_BENCH_CACHE", "/tmp"), verbose=False)
d, _ = bench.get_index(args, 0, 1, lambda: None)
d.to_device(0)
n, L = 20_000_000, 150
reads = make_reads_device(d, 0, n, L, positive_fraction=positive, seed=args.seed)
offsets = torch.arange(n + 1, dtype=torch.int64, device="cuda:0") * L
report = torch.zeros(16, dtype=torch.int64, device="cuda:0")
d.streaming_query_device(0, reads.data_ptr(), offsets.data_ptr(), n, report.data_ptr(), total_bases=n * L)
torch.cuda.synchronize()
v = [int(x) for x in report.cpu().tolist()]
names = ("wave_turns", "fresh_seeds", "walking", "runs", "slot1", "invalid_skips", "idle", "complete_path", "negatives_kept", "heavy_shortcuts")
st = dict(zip(names, v[6:]))
turns = max(1, st["wave_turns"])
print(os.path.basename(os.environ.get("SSHASH_AMD_LIBRARY", "shipped")), workload, positive, "report", v[:6])
print("  wave-turns", st["wave_turns"], "= %.2f per read" % (turns * 64 / n), "(if every lane were busy every turn)")
for k in names[1:8]:
    print(f"  lanes per turn {k:14s} {st[k] / turns:6.2f} of 64   (events per read: {st[k] / n:.2f})")
print("  fresh seeds that started on their own sequence (heavy key remembered): %.2f per read" % (st["heavy_shortcuts"] / n))
print("  negatives settled without a probe:", st["negatives_kept"], "= %.2f per fresh seed that missed" % (st["negatives_kept"] / max(1, st["fresh_seeds"] + st["walking"] - v[4])))
