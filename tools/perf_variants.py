#!/usr/bin/env python
"""Secondary measurements on the bench dictionary (not the headline number): entry-point variants,
query mixes, host-buffer (PCIe-inclusive) rate and the batched streaming query. Prints JSON lines.

    python tools/perf_variants.py [--bases B --queries Q --canonical --k K --m M]
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
    ap.add_argument("--bases", type=int, default=1_387_536_274)
    ap.add_argument("--queries", type=int, default=50_000_000)
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--m", type=int, default=21)
    ap.add_argument("--mean-len", type=float, default=85.0)
    ap.add_argument("--canonical", action="store_true")
    ap.add_argument("--seed", type=int, default=0x5555AAAA)
    ap.add_argument("--cache-dir", default="/tmp")
    ap.add_argument("--reads", type=int, default=2_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--skip-streaming", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()
    # the bench's own stand-ins (bench.get_index): the S. enterica recipe at C2 size by default, the human one for larger --bases
    args.recipe = "human_k63" if args.k > 31 else ("se_k31" if args.bases < 2_000_000_000 else "human_k31")
    args.repeat_scale = 1.0

    import torch

    import bench
    from sshash_amd.synthetic import draw_queries

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    d, path = bench.get_index(args, 0, 1, lambda: None)
    d.to_device(0)
    n, W, k = args.queries, d.words_per_kmer(), d.k()
    stream = torch.cuda.current_stream().cuda_stream
    base = {"k": k, "m": d.m(), "canonical": d.canonical(), "num_kmers": d.num_kmers(), "queries": n}

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    def report(name, ms, units=n, unit="lookups/s", **extra):
        print(json.dumps({"variant": name, "ms": round(ms, 3), "rate": round(units / ms * 1e3, 1), "unit": unit, **base, **extra}), flush=True)

    out = torch.empty(n, dtype=torch.int64, device=dev)
    for name, frac in [("mix50", 0.5), ("positive100", 1.0), ("negative100", 0.0)]:
        q = draw_queries(d, n, frac, seed=args.seed + 1)
        dq = torch.from_numpy(q.view(np.int64)).to(dev)
        report(f"packed_ids_{name}", timed(lambda: d.lookup_device(0, dq.data_ptr(), n, out.data_ptr(), stream=stream)))
        if name == "positive100":
            report("packed_ids_positive100_no_rc_check",
                   timed(lambda: d.lookup_device(0, dq.data_ptr(), n, out.data_ptr(), check_reverse_complement=False, stream=stream)))
    q = draw_queries(d, n, 0.5, seed=args.seed + 1)
    dq = torch.from_numpy(q.view(np.int64)).to(dev)
    mem = torch.empty(n, dtype=torch.uint8, device=dev)
    report("is_member_mix50", timed(lambda: d.is_member_device(0, dq.data_ptr(), n, mem.data_ptr(), stream=stream)))
    extra = {f: torch.empty(n, dtype=torch.int64, device=dev) for f in
             ("kmer_id_in_string", "kmer_offset", "string_id", "string_begin", "string_end")}
    ori = torch.empty(n, dtype=torch.int8, device=dev)
    mf = torch.empty(n, dtype=torch.uint8, device=dev)
    ptrs = {f: t.data_ptr() for f, t in extra.items()}
    ptrs.update(kmer_orientation=ori.data_ptr(), minimizer_found=mf.data_ptr())
    report("packed_full_result_mix50", timed(lambda: d.lookup_device(0, dq.data_ptr(), n, out.data_ptr(), stream=stream, **ptrs)))
    no_flag = {f: p for f, p in ptrs.items() if f != "minimizer_found"}
    report("packed_seven_fields_mix50", timed(lambda: d.lookup_device(0, dq.data_ptr(), n, out.data_ptr(), stream=stream, **no_flag)))
    out_other = torch.empty_like(out)  # (`out` keeps the ids of q: the host-buffer variants below are checked against it)
    for frac, name in ((1.0, "positive100"), (0.9, "positive90"), (0.0, "negative100")):
        dq2 = torch.from_numpy(draw_queries(d, n, frac, seed=args.seed + 5).view(np.int64)).to(dev)
        report("packed_full_result_" + name, timed(lambda: d.lookup_device(0, dq2.data_ptr(), n, out_other.data_ptr(), stream=stream, **ptrs)))
        del dq2
    del extra
    if W == 1:
        # ASCII form of the same batch (n*k bytes)
        codes = (q[:, None] >> (np.arange(k, dtype=np.uint64) * np.uint64(2))[None, :]) & np.uint64(3)
        ascii_q = np.frombuffer(b"ACTG", dtype=np.uint8)[codes.astype(np.uint8)]
        da = torch.from_numpy(np.ascontiguousarray(ascii_q)).to(dev)
        out2 = torch.empty(n, dtype=torch.int64, device=dev)
        report("ascii_ids_mix50", timed(lambda: d.lookup_device(0, da.data_ptr(), n, out2.data_ptr(), stream=stream, ascii_input=True)))
        d.lookup_device(0, dq.data_ptr(), n, out.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        assert torch.equal(out, out2), "ASCII and packed entry points disagree"
        del da, ascii_q, codes
    # host buffers (pageable), PCIe inclusive
    m = n
    t0 = time.perf_counter()
    ids = d.lookup(q[: m * W]).kmer_id
    t = time.perf_counter() - t0
    report("host_buffers_packed_mix50_pcie_inclusive_first_call", t * 1e3, units=m)  # creates the pinned lanes
    assert (ids == out[:m].cpu().numpy().view(np.uint64)).all()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        ids = d.lookup(q[: m * W]).kmer_id
        best = min(best, time.perf_counter() - t0)
    report("host_buffers_packed_mix50_pcie_inclusive", best * 1e3, units=m)
    assert (ids == out[:m].cpu().numpy().view(np.uint64)).all()
    # the same with page-locked caller buffers: copied from and to directly
    q_pin = torch.from_numpy(q[: m * W].view(np.int64)).pin_memory()
    ids_pin = torch.empty(m, dtype=torch.int64).pin_memory()
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        d.lookup(q_pin.numpy().view(np.uint64), out=ids_pin.numpy().view(np.uint64))
        best = min(best, time.perf_counter() - t0)
    report("host_buffers_page_locked_packed_mix50_pcie_inclusive", best * 1e3, units=m)
    assert (ids_pin.numpy().view(np.uint64) == ids).all()
    del q_pin, ids_pin

    if not args.skip_streaming:
        from oracle import oracle as O  # checker for the counters of a sample

        words, endpoints = bench.make_standin(args)  # the strings the index was built from (deterministic): reads are drawn from them
        total = int(endpoints[-1])
        rng = np.random.default_rng(99)
        L, R = args.read_len, args.reads
        lut = np.frombuffer(b"ACTG", dtype=np.uint8)
        comp = np.zeros(256, dtype=np.uint8)
        for a, b in zip(b"ACGT", b"TGCA"):
            comp[a] = b
        for name, hit in [("high_hit", True), ("low_hit", False)]:
            reads = np.empty((R, L), dtype=np.uint8)
            for a in range(0, R, 250_000):
                b = min(R, a + 250_000)
                if hit:
                    start = rng.integers(0, total - L, b - a)
                    pos = start[:, None] + np.arange(L)[None, :]
                    codes = ((words[pos >> 5] >> ((pos & 31).astype(np.uint64) * np.uint64(2))) & np.uint64(3)).astype(np.uint8)
                    r = lut[codes]
                    sub = rng.random((b - a, L)) < 0.01
                    r[sub] = lut[rng.integers(0, 4, int(sub.sum()))]
                    r[rng.random((b - a, L)) < 0.001] = ord("N")
                    flip = rng.random(b - a) < 0.5
                    r[flip] = comp[r[flip][:, ::-1]]
                    r[r == 0] = ord("N")
                    reads[a:b] = r
                else:
                    reads[a:b] = lut[rng.integers(0, 4, (b - a, L), dtype=np.uint8)]
            offsets = np.arange(R + 1, dtype=np.uint64) * np.uint64(L)
            d_bases = torch.from_numpy(reads.reshape(-1)).to(dev)
            d_off = torch.from_numpy(offsets.view(np.int64)).to(dev)
            rep = torch.zeros(6, dtype=torch.int64, device=dev)

            def run():
                rep.zero_()
                d.streaming_query_device(0, d_bases.data_ptr(), d_off.data_ptr(), R, rep.data_ptr(), stream=stream)

            ms = timed(run, reps=2)
            counters = rep.cpu().numpy().tolist()
            nk = R * (L - k + 1)
            names = ["num_kmers", "num_positive_kmers", "num_negative_kmers", "num_invalid_kmers", "num_searches", "num_extensions"]
            report(f"streaming_{name}", ms, units=nk, unit="k-mers/s", reads=R, read_len=L, **dict(zip(names, counters)))
            # the same reads with per-k-mer results (ids) out of the position-parallel pipeline
            ids = torch.empty(R * L, dtype=torch.int64, device=dev)
            rep2 = torch.zeros(6, dtype=torch.int64, device=dev)

            def run_lookup():
                rep2.zero_()
                d.streaming_lookup_device(0, d_bases.data_ptr(), d_off.data_ptr(), R, R * L, ids.data_ptr(), d_report=rep2.data_ptr(), stream=stream)

            ms2 = timed(run_lookup, reps=2)
            assert rep2.cpu().numpy().tolist() == counters, (rep2.cpu().numpy().tolist(), counters)
            report(f"streaming_lookup_per_kmer_results_{name}", ms2, units=nk, unit="k-mers/s", reads=R, read_len=L)
            del ids
            # parity of the counters on a sample of reads
            sample = [bytes(reads[i]) for i in range(0, R, max(1, R // 2000))]
            want = O.OracleIndex(path).streaming_query(sample)
            got = d.streaming_query(sample)
            assert want == {kk: getattr(got, kk) for kk in want}, (want, got)


if __name__ == "__main__":
    main()
