#!/usr/bin/env python
"""BASELINE.json configs 2 and 3 at full size on ONE MI355X (the multi-GPU forms shard queries / reads
over replicas, bench.py --gpus N): prints one JSON line per config.

    python tools/bench_configs.py c3     # human-scale k=31 index, ONE batch of 10^9 random queries
    python tools/bench_configs.py c4     # human-scale k=63 index, streaming_query over 10^8 reads of 150 bp

Synthetic data throughout (no network): the SPSS generator of sshash_amd/synthetic.py with human-genome
size statistics; reads = half sampled from the strings with 1 % substitutions and N at 10^-3, half random.
Ids / counters are checked against the CPU oracle on a sample.
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


def log(msg):
    print(f"[configs] {msg}", file=sys.stderr, flush=True)


def config_c3(args):
    import torch

    import bench
    from oracle import oracle as O
    from sshash_amd.synthetic import draw_queries

    ns = argparse.Namespace(bases=2_813_553_873, k=31, m=21, mean_len=274.0, canonical=False, seed=0x5555AAAA,
                            cache_dir=args.cache_dir, verbose=False)
    d, path = bench.get_index(ns, 0, 1, lambda: None)
    d.to_device(0)
    dev = torch.device("cuda", 0)
    n, chunk = args.queries, 100_000_000
    dq = torch.empty(n, dtype=torch.int64, device=dev)
    first = None
    for at in range(0, n, chunk):
        m = min(chunk, n - at)
        q = draw_queries(d, m, 0.5, seed=ns.seed + 17 * (at // chunk))
        if first is None:
            first = q[:200_000].copy()
        dq[at:at + m] = torch.from_numpy(q.view(np.int64)).to(dev)
        log(f"queries {at + m}/{n}")
    out = torch.empty(n, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    run = lambda: d.lookup_device(0, dq.data_ptr(), n, out.data_ptr(), stream=stream)  # noqa: E731
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.reps
    want = O.OracleIndex(path).lookup_ids(first, num_threads=bench.effective_cores())
    got = out[:200_000].cpu().numpy().view(np.uint64)
    if not (got == want).all():
        raise SystemExit("PARITY FAILURE (c3)")
    found = float((out != -1).float().mean().item())
    print(json.dumps({"config": "C3: human-scale k=31 m=21 regular, one batch of %d random queries (50%% positive) on 1 MI355X, index replicated" % n,
                      "num_kmers": d.num_kmers(), "queries": n, "ms": round(ms, 2), "lookups_per_s": round(n / ms * 1e3, 1),
                      "fraction_found": round(found, 4), "device_index_bytes": d.device_bytes(0),
                      "parity": "200000 ids equal to the CPU oracle"}), flush=True)


def config_c4(args):
    import torch

    import bench
    from oracle import oracle as O
    from sshash_amd.synthetic import make_spss

    ns = argparse.Namespace(bases=3_000_000_000, k=63, m=25, mean_len=300.0, canonical=False, seed=0x5555AAAA,
                            cache_dir=args.cache_dir, verbose=False)
    d, path = bench.get_index(ns, 0, 1, lambda: None)
    d.to_device(0)
    dev = torch.device("cuda", 0)
    words, endpoints = make_spss(ns.bases, k=ns.k, m=ns.m, seed=ns.seed, mean_len=ns.mean_len)
    total = int(endpoints[-1])
    rng = np.random.default_rng(4)
    L, R, tiles = 150, args.reads // args.tiles, args.tiles
    lut = np.frombuffer(b"ACTG", dtype=np.uint8)
    comp = np.zeros(256, dtype=np.uint8)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    reads = np.empty((R, L), dtype=np.uint8)
    t0 = time.time()
    for a in range(0, R, 250_000):
        b = min(R, a + 250_000)
        half = (b - a) // 2
        start = rng.integers(0, total - L, half)
        pos = start[:, None] + np.arange(L)[None, :]
        codes = ((words[pos >> 5] >> ((pos & 31).astype(np.uint64) * np.uint64(2))) & np.uint64(3)).astype(np.uint8)
        r = lut[codes]
        sub = rng.random((half, L)) < 0.01
        r[sub] = lut[rng.integers(0, 4, int(sub.sum()))]
        flip = rng.random(half) < 0.5
        r[flip] = comp[r[flip][:, ::-1]]
        reads[a:a + half] = r
        reads[a + half:b] = lut[rng.integers(0, 4, (b - a - half, L), dtype=np.uint8)]
        blk = reads[a:b]
        blk[rng.random((b - a, L)) < 0.001] = ord("N")
    del words
    reads = reads[rng.permutation(R)]
    log(f"{R} reads generated in {time.time() - t0:.0f}s; tiled x{tiles} on the device")
    one = torch.from_numpy(reads.reshape(-1)).to(dev)
    d_bases = one.repeat(tiles)
    del one
    n_reads = R * tiles
    d_off = torch.arange(n_reads + 1, dtype=torch.int64, device=dev) * L
    rep = torch.zeros(6, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def run():
        rep.zero_()
        d.streaming_query_device(0, d_bases.data_ptr(), d_off.data_ptr(), n_reads, rep.data_ptr(), stream=stream)

    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.reps
    counters = [int(x) for x in rep.cpu().tolist()]
    # the oracle on a sample of reads, the device on the same sample
    S = 20_000
    sample = [bytes(r) for r in reads[:S]]
    want = O.OracleIndex(path).streaming_query(sample)
    got = d.streaming_query(sample)
    names = ["num_kmers", "num_positive_kmers", "num_negative_kmers", "num_invalid_kmers", "num_searches", "num_extensions"]
    got_l = [int(got[x]) if isinstance(got, dict) else int(getattr(got, x)) for x in names]
    want_l = [int(want[x]) if isinstance(want, dict) else int(getattr(want, x)) for x in names]
    if got_l != want_l:
        raise SystemExit(f"PARITY FAILURE (c4): {got_l} != {want_l}")
    kmers = counters[0]
    print(json.dumps({"config": "C4: human-scale k=63 m=25 regular, streaming_query over %d reads of %d bp on 1 MI355X (%d distinct reads tiled x%d)" % (n_reads, L, R, tiles),
                      "num_kmers_in_index": d.num_kmers(), "reads": n_reads, "ms": round(ms, 2),
                      "kmers_per_s": round(kmers / ms * 1e3, 1), "reads_per_s": round(n_reads / ms * 1e3, 1),
                      "report": dict(zip(names, counters)), "device_index_bytes": d.device_bytes(0),
                      "parity": "six counters of a %d-read sample equal to the CPU oracle" % S}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config", choices=["c3", "c4"])
    ap.add_argument("--queries", type=int, default=1_000_000_000)
    ap.add_argument("--reads", type=int, default=100_000_000)
    ap.add_argument("--tiles", type=int, default=10)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--cache-dir", default="/tmp")
    args = ap.parse_args()
    (config_c3 if args.config == "c3" else config_c4)(args)


if __name__ == "__main__":
    main()
