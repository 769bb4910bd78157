#!/usr/bin/env python
"""End-to-end `sshash query` on a FASTQ file: sshash_streaming_query_from_file, wall clock INCLUDING decompression and parsing,
reported like the reference's tools/query.cpp:38-67 (num_kmers, elapsed, ns/k-mer), next to the published 89.5 ns/k-mer
(benchmarks/results-21-01-26/k31/regular-streaming-queries-high-hit.json:3: human k=31, SRR5833294, one 5.4 GHz core, gzipped).

    python tools/bench_streaming_file.py [--reads 100000000] [--bases 2813192630] [--dir /tmp]

Prints one JSON line: per file flavour (plain .fastq, .fastq.gz) the wall time, k-mers/s, ns/k-mer and the six counters; the
reader alone (tools/reader_rate: decompress + split, no GPU) and the CPU oracle's streaming query on a bounded sample of the
same reads beside it."""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=100_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--positive", type=float, default=0.9)
    ap.add_argument("--bases", type=int, default=2_813_192_630)
    ap.add_argument("--dir", default="/tmp")
    ap.add_argument("--c4", action="store_true",
                    help="BASELINE config C4: the human k = 63, m = 25 stand-in (recipe human_k63), half of the reads drawn from it, half random")
    args = ap.parse_args()
    if args.c4:
        args.positive = 0.5
        if args.bases == 2_813_192_630:
            args.bases = 2_935_176_947
    import argparse as A

    import bench

    b = A.Namespace(bases=args.bases, k=63 if args.c4 else 31, m=25 if args.c4 else 21, canonical=False, seed=0x5555AAAA, cache_dir=args.dir,
                    verbose=False, recipe="human_k63" if args.c4 else "human_k31", repeat_scale=1.0)
    d, index_path = bench.get_index(b, 0, 1, lambda: None)
    d.to_device(0)
    from sshash_amd.synthetic import make_reads_device

    t0 = time.time()
    reads = make_reads_device(d, 0, args.reads, args.read_len, positive_fraction=args.positive).cpu()
    print(f"[file] {args.reads} reads drawn in {time.time() - t0:.1f}s", file=sys.stderr, flush=True)
    res = bench.measure_streaming_from_file(d, index_path, reads, args.dir, "reads", log=lambda *a: print("[file]", *a, file=sys.stderr, flush=True))
    res["workload"] = f"{args.reads} reads x {args.read_len} bp, {args.positive:.0%} drawn from the dictionary (1 % substitutions), N at 1e-3; " + ("C4 stand-in (k = 63)" if args.c4 else "C3 stand-in")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
