#!/usr/bin/env python
"""bench.py -- batched random k-mer Lookup throughput on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (sshash_lookup_packed_device: packed k-mers in HBM ->
k-mer ids in HBM) over one batch of synthetic queries per GPU. Workload = BASELINE.json
configs[1]: an index of the size statistics of the S. enterica pangenome (k=31, m=21; the real
collection is not available offline, so a synthetic stand-in is generated -- sshash_amd/synthetic.py),
100 M queries per batch, 50 % positive (half of those reverse-complemented), 50 % uniform random,
shuffled, seeded.

    python bench.py [--gpus N --steps K --warmup W] [--bases B --queries Q] [--canonical]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: one process per GPU, the index replicated in each HBM, every rank looks up its own batch
(weak scaling); no collective on the data path -- torch.distributed (RCCL) is used only for the
barriers around the timed region and the max-over-ranks of the elapsed time.

Rank 0 prints ONE JSON line (fields: see the task contract; `roofline` and `cpu_baseline` added).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def effective_cores() -> int:
    """CPUs this process may actually use: min(affinity, cgroup quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def get_index(args, rank: int, world: int, barrier):
    """Build the synthetic dictionary once (rank 0), cache it on local disk, load it on every rank."""
    import sshash_amd
    from sshash_amd.synthetic import make_spss

    mean_len = getattr(args, "mean_len", 85.0)
    key = f"v3-{args.bases}-{args.k}-{args.m}-{int(args.canonical)}-{args.seed}-{mean_len}"
    path = os.path.join(args.cache_dir, "sshash_amd_bench_" + hashlib.sha1(key.encode()).hexdigest()[:16] + ".sshash")
    if rank == 0 and not os.path.exists(path):
        t0 = time.time()
        words, endpoints = make_spss(args.bases, k=args.k, m=args.m, seed=args.seed, mean_len=mean_len)
        log(f"synthetic SPSS: {endpoints.size - 1} strings, {int(endpoints[-1])} bases in {time.time() - t0:.1f}s")
        t0 = time.time()
        d = sshash_amd.Dictionary.build_from_packed(words, endpoints, k=args.k, m=args.m, canonical=args.canonical,
                                                    num_threads=0, verbose=args.verbose)
        log(f"dictionary built in {time.time() - t0:.1f}s: {d.num_kmers()} k-mers, {d.num_minimizers()} minimizers, "
            f"{d.num_bits() / 8e6:.0f} MB ({d.num_bits() / d.num_kmers():.2f} bits/k-mer)")
        tmp = path + f".tmp{os.getpid()}"
        d.save(tmp)
        os.replace(tmp, path)
        if world == 1:
            return d, path
        d.close()
    barrier()
    t0 = time.time()
    d = sshash_amd.Dictionary.load(path)
    if rank == 0:
        log(f"dictionary loaded from cache in {time.time() - t0:.1f}s")
    return d, path


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--bases", type=int, default=1_387_536_274, help="bases in the synthetic SPSS (C2: S. enterica pangenome)")
    ap.add_argument("--queries", type=int, default=100_000_000, help="queries per GPU per step")
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--m", type=int, default=21)
    ap.add_argument("--mean-len", type=float, default=85.0, help="mean string length of the synthetic SPSS")
    ap.add_argument("--canonical", action="store_true")
    ap.add_argument("--seed", type=int, default=0x5555AAAA)
    ap.add_argument("--cache-dir", default=os.environ.get("SSHASH_BENCH_CACHE", "/tmp"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=2_000_000, help="queries per CPU-baseline pass")
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the lookup path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # under torch.distributed.run (RANK set) the RCCL group is always created, also for one rank, so
    # that the very same code path runs at N = 1, 2, 4, 8
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend="nccl", device_id=dev, rank=rank, world_size=world)

    def barrier():
        if use_dist:
            dist.barrier()

    from sshash_amd.synthetic import draw_queries

    d, index_path = get_index(args, rank, world, barrier)
    t0 = time.time()
    d.to_device(local_rank)
    if rank == 0:
        log(f"replica in HBM: {d.device_bytes(local_rank) / 1e6:.0f} MB (upload {time.time() - t0:.1f}s); {d.device_stats(local_rank)}")

    n = args.queries
    W = d.words_per_kmer()
    t0 = time.time()
    queries = draw_queries(d, n, 0.5, seed=args.seed + 7919 * rank)
    if rank == 0:
        log(f"{n} queries drawn in {time.time() - t0:.1f}s")
    dq = torch.from_numpy(queries.view(np.int64)).to(dev)
    out = torch.empty(n, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream()

    def step():
        d.lookup_device(local_rank, dq.data_ptr(), n, out.data_ptr(), check_reverse_complement=True, stream=stream.cuda_stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    stops = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    t_begin = time.perf_counter()
    for i in range(args.steps):
        starts[i].record(stream)
        step()
        stops[i].record(stream)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_begin
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = [starts[i].elapsed_time(stops[i]) for i in range(args.steps)]
    avg_kernel_ms = float(np.mean(kernel_ms))

    # ---- parity spot check + algorithmic bytes (oracle = checker only) ---------------------------
    result = None
    if rank == 0:
        from oracle import oracle as O

        ora = O.OracleIndex(index_path)
        sample = 200_000
        got = out[:sample].cpu().numpy().view(np.uint64)
        want = ora.lookup_ids(queries[: sample * W], num_threads=effective_cores())
        if not (got == want).all():
            raise SystemExit("PARITY FAILURE: GPU ids differ from the CPU oracle")
        found = float((got != np.uint64(0xFFFFFFFFFFFFFFFF)).mean())
        bytes_per_lookup = ora.count_bytes(queries[: 100_000 * W]) / 100_000
        achieved = bytes_per_lookup * n / (avg_kernel_ms * 1e-3) / 1e9
        traffic = None
        prof = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(prof):
            try:
                rec = json.load(open(prof))
                if rec.get("queries") == n and rec.get("bases") == args.bases and rec.get("canonical") == args.canonical:
                    traffic = rec.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "kernel": "fast_lookup_kernel<W=%d,canonical=%d,ids,%s> + deferred_lookup_kernel (one pair per step; "
                              "avg_kernel_ms = HIP-event time around the pair)"
                              % (W, int(d.canonical()), "super-k-mer table" if d.device_stats(local_rank)["sk_slots"] else "directory"),
                    "algorithmic_bytes_per_lookup": round(bytes_per_lookup, 2), "avg_kernel_ms": round(avg_kernel_ms, 3)}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cores = effective_cores()
            m = min(args.cpu_sample, n)
            q1 = queries[: m * W]
            t0 = time.perf_counter()
            ora.lookup_ids(q1, num_threads=1)
            t1 = time.perf_counter() - t0
            reps = max(1, min(64, int(20.0 / max(1e-3, t1 / cores * 4))))
            big = min(n, m * reps)
            qa = queries[: big * W]
            t0 = time.perf_counter()
            ids_all = ora.lookup_ids(qa, num_threads=cores)
            ta = time.perf_counter() - t0
            assert (ids_all[: min(big, n)] == out[:big].cpu().numpy().view(np.uint64)).all()
            cpu = {"value": round(big / ta, 1), "unit": "lookups/s", "cores": cores, "kind": "port",
                   "sample": f"{big} queries of the same batch on {cores} threads (= usable CPUs: affinity {os.cpu_count()}, "
                             f"cgroup quota applied; contiguous chunks); "
                             f"1 thread: {m / t1:.0f} lookups/s = {t1 / m * 1e9:.0f} ns/lookup over {m} queries",
                   "single_thread_value": round(m / t1, 1)}
        total = n * world * args.steps
        result = {
            "metric": "k-mer Lookups/sec (batched random queries, bit-exact ids)",
            "value": round(total / elapsed, 1),
            "unit": "lookups/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": "C2 stand-in: synthetic SPSS with S. enterica pangenome size statistics, "
                                   f"k={d.k()} m={d.m()} {'canonical' if d.canonical() else 'regular'}, "
                                   f"{d.num_kmers()} k-mers / {d.num_strings()} strings / {d.num_bases()} bases, "
                                   f"{n} packed queries per GPU per step (50% positive, half of them reverse-complemented)",
                       "queries_per_gpu": n, "num_kmers": d.num_kmers(), "k": d.k(), "m": d.m(),
                       "canonical": d.canonical(), "index_replicated_per_gpu": True,
                       "positive_fraction_found": round(found, 4), "device_index_bytes": d.device_bytes(local_rank)},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
    barrier()
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
