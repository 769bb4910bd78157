#!/usr/bin/env python
"""Cross-check of the layered lookup structures on the full-size bench dictionary: the ids of one 10^8-query
batch through the super-k-mer table must equal those through directory + atoms (SSHASH_AMD_SKTABLE=0) and
through the bare MPHF path (also SSHASH_AMD_DIRECTORY=0). Each variant runs in its own process (the switches are
read at upload). Prints one JSON line.

    python tools/crosscheck_paths.py [--k 31 --m 21 --bases B --queries Q --canonical]
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(args):
    import numpy as np
    import torch

    import bench
    from sshash_amd.synthetic import draw_queries

    d, _ = bench.get_index(args, 0, 1, lambda: None)
    d.to_device(0)
    q = draw_queries(d, args.queries, 0.5, seed=args.seed + 3)
    dq = torch.from_numpy(q.view(np.int64)).cuda()
    n = args.queries
    out = torch.empty(n, dtype=torch.int64, device="cuda")
    mem = torch.empty(n, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    d.lookup_device(0, dq.data_ptr(), n, out.data_ptr(), stream=s)
    d.is_member_device(0, dq.data_ptr(), n, mem.data_ptr(), stream=s)
    torch.cuda.synchronize()
    ids = out.cpu().numpy()
    assert ((ids != -1) == mem.cpu().numpy().astype(bool)).all(), "is_member disagrees with lookup"
    np.save(args.out, ids)
    print(json.dumps(d.device_stats(0)), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bases", type=int, default=1_387_536_274)
    ap.add_argument("--queries", type=int, default=100_000_000)
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--m", type=int, default=21)
    ap.add_argument("--mean-len", type=float, default=85.0)
    ap.add_argument("--canonical", action="store_true")
    ap.add_argument("--seed", type=int, default=0x5555AAAA)
    ap.add_argument("--cache-dir", default="/tmp")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--out", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.out:
        return worker(args)
    import numpy as np

    variants = {"table": {}, "directory": {"SSHASH_AMD_SKTABLE": "0"},
                "mphf": {"SSHASH_AMD_SKTABLE": "0", "SSHASH_AMD_DIRECTORY": "0"}}
    with tempfile.TemporaryDirectory() as tmp:
        stats = {}
        for name, env in variants.items():
            path = os.path.join(tmp, name + ".npy")
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--out", path] + sys.argv[1:],
                               env=dict(os.environ, **env), capture_output=True, text=True)
            if p.returncode != 0:
                raise SystemExit(f"{name}: {p.stdout}{p.stderr}")
            stats[name] = json.loads(p.stdout.strip().splitlines()[-1])
        ref = np.load(os.path.join(tmp, "mphf.npy"))
        result = {"k": args.k, "m": args.m, "canonical": args.canonical, "queries": int(ref.size), "found": int((ref != -1).sum())}
        for name in ("table", "directory"):
            got = np.load(os.path.join(tmp, name + ".npy"))
            result[name + "_mismatches_vs_mphf"] = int((got != ref).sum())
        result["sk_slots"] = stats["table"]["sk_slots"]
        result["sk_deferred_keys"] = stats["table"]["sk_deferred_keys"]
    print(json.dumps(result), flush=True)
    if result["table_mismatches_vs_mphf"] or result["directory_mismatches_vs_mphf"]:
        raise SystemExit("MISMATCH between lookup paths")


if __name__ == "__main__":
    main()
