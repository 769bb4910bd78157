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

import numpy as np  # noqa: E402


def measure(d, index_path, reads_tensor, directory, tag, oracle_sample=100_000, gzip_level=1, log=lambda *a: None):
    import sshash_amd
    from sshash_amd.synthetic import write_fastq

    k = d.k()
    n, L = reads_tensor.shape
    res = {"reads": int(n), "read_length": int(L), "kmers": int(n) * (L - k + 1)}
    exe = os.path.join(directory, "reader_rate")
    have_exe = subprocess.call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tools", "reader_rate.cpp"),
                                os.path.join(ROOT, "sshash_amd", "csrc", "reads.cpp"), "-lz", "-o", exe]) == 0
    for flavour, level in (("fastq", None), ("fastq.gz", gzip_level)):
        path = os.path.join(directory, f"sshash_amd_{tag}.{flavour}")
        t0 = time.perf_counter()
        size = write_fastq(reads_tensor, path, gzip_level=level, workers=max(1, (os.cpu_count() or 8) // 2))
        log(f"{path}: {size / 1e9:.2f} GB written in {time.perf_counter() - t0:.1f}s")
        d.streaming_query_from_file(path)  # (page cache warm, pools sized: the reference's numbers are warm-cache too)
        t0 = time.perf_counter()
        rep = d.streaming_query_from_file(path)
        dt = time.perf_counter() - t0
        entry = {"file_bytes": size, "seconds": round(dt, 3), "kmers_per_s": round(rep.num_kmers / dt, 1),
                 "ns_per_kmer": round(dt / max(1, rep.num_kmers) * 1e9, 3),
                 "report": {f: int(getattr(rep, f)) for f in ("num_kmers", "num_positive_kmers", "num_negative_kmers", "num_invalid_kmers",
                                                              "num_searches", "num_extensions")}}
        if have_exe:
            out = subprocess.run([exe, path, str(k)], capture_output=True, text=True)
            if out.returncode == 0:
                entry["reader_alone"] = json.loads(out.stdout)
        res[flavour] = entry
        os.remove(path)
    # the CPU oracle's streaming state machine on the first reads of the same file (1 thread, as the reference's query tool)
    from oracle import oracle as O

    ora = O.OracleIndex(index_path)
    m = min(n, oracle_sample)
    sample = reads_tensor[:m].cpu().numpy()
    reads = [bytes(r) for r in sample]
    t0 = time.perf_counter()
    want = ora.streaming_query(reads)
    dt = time.perf_counter() - t0
    res["cpu_oracle"] = {"kind": "port", "cores": 1, "reads": m, "seconds": round(dt, 3), "ns_per_kmer": round(dt / max(1, want["num_kmers"]) * 1e9, 2),
                         "kmers_per_s": round(want["num_kmers"] / dt, 1)}
    got = d.streaming_query(reads)
    for f, v in want.items():
        if int(getattr(got, f)) != v:
            raise SystemExit(f"PARITY FAILURE: streaming counter {f}: GPU {getattr(got, f)} vs oracle {v}")
    res["counters_equal_oracle_on_sample"] = True
    res["published_reference"] = {"ns_per_kmer": 89.5, "what": "human k=31 regular, SRR5833294 (91.6 % positive), gzipped FASTQ, one 5.4 GHz core, "
                                  "benchmarks/results-21-01-26/k31/regular-streaming-queries-high-hit.json:3"}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=100_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--positive", type=float, default=0.9)
    ap.add_argument("--bases", type=int, default=2_813_192_630)
    ap.add_argument("--dir", default="/tmp")
    args = ap.parse_args()
    import argparse as A

    import bench

    b = A.Namespace(bases=args.bases, k=31, m=21, canonical=False, seed=0x5555AAAA, cache_dir=args.dir, verbose=False, recipe="human_k31",
                    repeat_scale=1.0)
    d, index_path = bench.get_index(b, 0, 1, lambda: None)
    d.to_device(0)
    from sshash_amd.synthetic import make_reads_device

    t0 = time.time()
    reads = make_reads_device(d, 0, args.reads, args.read_len, positive_fraction=args.positive).cpu()
    print(f"[file] {args.reads} reads drawn in {time.time() - t0:.1f}s", file=sys.stderr, flush=True)
    res = measure(d, index_path, reads, args.dir, "reads", log=lambda *a: print("[file]", *a, file=sys.stderr, flush=True))
    res["workload"] = f"{args.reads} reads x {args.read_len} bp, {args.positive:.0%} drawn from the dictionary (1 % substitutions), N at 1e-3; C3 stand-in"
    print(json.dumps(res))


if __name__ == "__main__":
    main()
