#!/usr/bin/env python
"""Throughput of the minimizer-sharded mode (sshash_amd/sharded.py; BASELINE.json config 5).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/bench_sharded.py \\
           [--bases B --queries Q --steps K]

Every rank builds ITS shard of the synthetic dictionary (strings complete, minimizer structures 1/N),
draws its own query batch and runs K routed lookups (route -> all_to_all -> lookup -> all_to_all ->
combine) over RCCL. Rank 0 prints one JSON line; ids of a sample are checked against the unsharded
dictionary's CPU oracle when --check is given (needs host memory for the whole index).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bases", type=int, default=200_000_000)
    ap.add_argument("--queries", type=int, default=20_000_000, help="queries per rank per step")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--m", type=int, default=21)
    ap.add_argument("--canonical", action="store_true")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--by", choices=["minimizer", "table"], default="minimizer",
                    help="minimizer: rank r holds its minimizer shard of the index; table: every rank holds the complete "
                         "index and 1/N of the device's super-k-mer table, one message per query")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import sshash_amd
    from sshash_amd.sharded import ShardedDictionary
    from sshash_amd.synthetic import draw_queries, make_spss

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29512")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group(backend="nccl", device_id=dev, rank=rank, world_size=world)
    words, ends = make_spss(args.bases, k=args.k, m=args.m)
    t0 = time.time()
    parts = (world, rank) if args.by == "minimizer" else (1, 0)
    shard = sshash_amd.Dictionary.build_from_packed(words, ends, k=args.k, m=args.m, canonical=args.canonical, num_threads=0,
                                                    num_shards=parts[0], shard_id=parts[1])
    sd = ShardedDictionary(shard, local, by=args.by)
    build_s = time.time() - t0
    q = draw_queries(shard, args.queries, 0.5, seed=1234 + rank)
    dq = torch.from_numpy(q.view(np.int64)).to(dev)
    for _ in range(args.warmup):
        out = sd.lookup_device(dq)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = sd.lookup_device(dq)
    torch.cuda.synchronize()
    dist.barrier()
    elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    found = (out != -1).sum().to(torch.int64)
    dist.all_reduce(found)
    ok = None
    if args.check and rank == 0:
        import tempfile

        from oracle import oracle as O

        whole = sshash_amd.Dictionary.build_from_packed(words, ends, k=args.k, m=args.m, canonical=args.canonical, num_threads=0)
        with tempfile.TemporaryDirectory() as tmp:
            p = os.path.join(tmp, "whole.sshash")
            whole.save(p)
            want = O.OracleIndex(p).lookup_ids(q[: 100_000 * shard.words_per_kmer()], num_threads=8)
        ok = bool((out[:100_000].cpu().numpy().view(np.uint64) == want).all())
        if not ok:
            raise SystemExit("PARITY FAILURE in sharded mode")
    if rank == 0:
        print(json.dumps({"metric": "k-mer Lookups/sec, %s-sharded index with all-to-all routing" % args.by, "n_gpus": world,
                          "value": round(args.queries * world * args.steps / float(elapsed), 1), "unit": "lookups/s",
                          "ms_per_step": round(float(elapsed) / args.steps * 1e3, 3), "queries_per_gpu": args.queries,
                          "num_kmers": shard.num_kmers(), "shard_minimizers": shard.num_minimizers(),
                          "shard_device_bytes": shard.device_bytes(local), "shard_build_s": round(build_s, 1),
                          "fraction_found": round(int(found) / (args.queries * world), 4), "parity_checked": ok}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
