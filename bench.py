#!/usr/bin/env python
"""bench.py -- batched random k-mer Lookup throughput on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (sshash_lookup_packed_device: packed k-mers in HBM -> k-mer ids in
HBM) over ONE batch of synthetic queries, split over the GPUs. Default workload = BASELINE.json configs[2]
(C3, the configuration the metric and the >= 1e9 lookups/s target are quoted on): a human-genome-scale k=31
m=21 index (the real unitigs are not available offline, so a synthetic stand-in with the human build's size
statistics is generated -- sshash_amd/synthetic.py; reference benchmarks/results-10-11-25/k31/regular-build.json:3)
replicated in every GPU's HBM, and one batch of 10^9 random queries -- 50 % positive (half of those
reverse-complemented), 50 % uniform random (tools/perf.hpp:38-74), seeded, drawn on the device -- sharded over
the ranks ("scaling": "strong"). `--workload c2` selects configs[1] (S. enterica pangenome scale, 10^8 queries).

    python bench.py [--gpus N --steps K --warmup W] [--workload c3|c2|c4] [--bases B --queries Q] [--canonical]
    python bench.py --workload c4 --streaming [--gpus N --reads R]      the streaming query of configs[3], read-sharded

The default run (C3, one GPU) appends `other_workloads`: the C2 line, the C4 (k = 63) line, the C4 streaming line and the k = 31 streaming line, each a
child run of this script with its own oracle check, roofline and cpu_baseline.

`--gpus N` with N > 1 starts the N ranks itself (python -m torch.distributed.run, rendezvous on 127.0.0.1);
launched under torch.distributed.run it uses the ranks it is given. One process per GPU; no collective on the
data path -- torch.distributed (RCCL) carries only the barriers around the timed region and the reductions of
the elapsed times.

Rank 0 prints ONE JSON line (fields: the task contract; `roofline`, `cpu_baseline`, `per_rank` added).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

RANDOM_UNIT_PROBE = 43.75e9  # random 64-byte units/s one MI355X sustains on a 32 GiB array (tools/tlb_probe)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md

WORKLOADS = {
    # name: (bases, recipe of sshash_amd/recipes/, queries in the batch, description)
    "c3": (2_813_192_630, "human_k31", 1_000_000_000,
           "C3 stand-in: synthetic SPSS fitted to the bucket statistics the reference printed for human.k31 (repeat families + "
           "de-duplication, sshash_amd/repeats.py; target vs achieved in config.index_statistics)"),
    "c4": (2_935_176_947, "human_k63", 1_000_000_000,
           "C4-scale dictionary (BASELINE.json configs[3] is its streaming query): synthetic SPSS fitted to the bucket statistics the reference "
           "printed for human.k63 (k=63 m=25: two-word k-mers)"),
    "c2": (1_387_536_274, "se_k31", 100_000_000,
           "C2 stand-in: synthetic SPSS fitted to the bucket statistics the reference printed for the S. enterica pangenome "
           "(sshash_amd/repeats.py; target vs achieved in config.index_statistics)"),
}


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


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def spawn_ranks(n: int, script: str, argv: list[str]) -> int:
    """`python bench.py --gpus N ...` outside torch.distributed.run: start the N ranks on this node."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), script] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, effective_cores() // n)))
    return subprocess.call(cmd, env=env)


# What "ids equal the oracle" does and does not rest on: said where the numbers are.
PARITY_NOTE = {"pinned": "ids, orientation, string fields (reference-compiled primitive vectors + the reference's input-order id contract on its own data files)",
               "unpinned": "minimizer_found of a miss, the searches/extensions split of the streaming report, MPHF values (PTHash sources absent from the "
                           "reference checkout: restatement only; tests/test_reference_index.py is the recipe that pins them)"}


def split_batch(total: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous share [lo, hi) of a batch of `total` queries for `rank` (tests/test_multiproc_gloo.py)."""
    return total * rank // world, total * (rank + 1) // world


def get_index(args, rank: int, world: int, barrier):
    """Build the synthetic dictionary once (rank 0), cache it on local disk, load it on every rank."""
    import sshash_amd

    key = f"v5-{args.bases}-{args.k}-{args.m}-{int(args.canonical)}-{args.seed}-{args.recipe}-{args.repeat_scale}-{recipe_digest(args.recipe)}"
    path = os.path.join(args.cache_dir, "sshash_amd_bench_" + hashlib.sha1(key.encode()).hexdigest()[:16] + ".sshash")
    if rank == 0 and not os.path.exists(path):
        t0 = time.time()
        words, endpoints = make_standin(args)
        log(f"synthetic SPSS: {endpoints.size - 1} strings, {int(endpoints[-1])} bases in {time.time() - t0:.1f}s")
        t0 = time.time()
        d = sshash_amd.Dictionary.build_from_packed(words, endpoints, k=args.k, m=args.m, canonical=args.canonical,
                                                    num_threads=0, verbose=args.verbose)
        del words
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


def recipe_digest(name: str) -> str:
    path = os.path.join(ROOT, "sshash_amd", "recipes", name + ".json")
    return hashlib.sha1(open(path, "rb").read()).hexdigest()[:12]


def make_standin(args):
    """The synthetic SPSS of this run: the named recipe at --bases bases; --repeat-scale multiplies the amount of every repeat
    family (the heavy-key sweep of HISTORY.md; 1.0 = the fitted recipe)."""
    import torch
    from sshash_amd.repeats import load_recipe, make_repeat_spss

    r = load_recipe(args.recipe)
    if (args.k > 31) != (int(r["k"]) > 31):  # (no recipe for this k-mer width: the planted-motif generator of rounds 1-2)
        from sshash_amd.synthetic import make_spss

        return make_spss(args.bases, k=args.k, m=args.m, seed=args.seed, mean_len=274.0)
    classes = [dict(c, families=c["families"] * args.repeat_scale) for c in r["classes"]]
    background = r["background"]
    if args.repeat_scale != 1.0:
        background = None  # the background fills whatever the families leave of --bases
    mean_len = max([b["mean_len"] for b in r["background"]] + [400.0])
    out = make_repeat_spss(args.bases, k=args.k, classes=classes, seed=args.seed, reference_bases=float(r["reference_bases"]),
                           background=background, mean_len=mean_len)
    if torch.cuda.is_available():
        torch.cuda.empty_cache()  # the generator's scratch must not count as used HBM when the table is sized
    return out


def measure_other_paths(args, index_path, device, d, dq, W, bytes_per_lookup):
    """Side measurement, OUTSIDE the timed region: the same batch (its first 10^8 queries) through the replica layouts that
    have no super-k-mer table -- what a minimizer shard, a dictionary too large for a table, or SSHASH_AMD_SKTABLE=0 run:
      directory   minimizer -> one-atom directory -> bucket probe (lookup_device.hpp fast_probe_*)
      mphf        the path north_star names: minimizer -> MPHF -> control codeword -> bucket probe
    Every id is compared with the table path's (itself checked against the oracle)."""
    import torch

    import sshash_amd

    m = min(dq.numel() // W, 100_000_000)
    out = torch.empty(m, dtype=torch.int64, device=dq.device)
    stream = torch.cuda.current_stream()
    ids_of_table_path = torch.empty(m, dtype=torch.int64, device=dq.device)
    d.lookup_device(device, dq.data_ptr(), m, ids_of_table_path.data_ptr(), check_reverse_complement=True, stream=stream.cuda_stream)
    res = {}
    for name, env in (("directory", {"SSHASH_AMD_SKTABLE": "0", "SSHASH_AMD_DIRECTORY": "1"}), ("mphf", {"SSHASH_AMD_SKTABLE": "0", "SSHASH_AMD_DIRECTORY": "0"})):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            d2 = sshash_amd.Dictionary.load(index_path)
            d2.to_device(device)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        st = d2.device_stats(device)
        assert st["sk_slots"] == 0 and (name == "directory") == bool(st["directory_sectors"]), st

        def run():
            d2.lookup_device(device, dq.data_ptr(), m, out.data_ptr(), check_reverse_complement=True, stream=stream.cuda_stream)

        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(3):
            run()
        e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        if not bool((out == ids_of_table_path).all().item()):
            raise SystemExit(f"PARITY FAILURE: the {name} path and the table path disagree")
        gbs = bytes_per_lookup * m / (ms * 1e-3) / 1e9
        traffic, provenance = traffic_record(d, m, args, path=name)
        res[name] = {"lookups_per_s": round(m / ms * 1e3, 1), "ms": round(ms, 3), "queries": m, "ids_equal_table_path": True,
                     "device_index_bytes": st["bytes"], "roofline_frac": round(gbs / HBM_PEAK_GBS, 5), "algorithmic_GBps": round(gbs, 1),
                     "traffic": traffic, "traffic_provenance": provenance,
                     "hbm_traffic_bytes_per_lookup": round(traffic / m, 2) if traffic else None,
                     "frac_hbm_traffic": round(traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if traffic else None}
        d2.close()
    res.update(measure_host_entry_points(d, device, dq, W, ids_of_table_path))
    return res


def measure_host_entry_points(d, device, dq, W, ids_of_device_path, queries=50_000_000):
    """Side measurement, OUTSIDE the timed region, PCIe INCLUSIVE (never `value`): the entry points a drop-in caller of the reference's
    `dict.lookup(...)` loop uses (tools/perf.hpp:57,79) -- sshash_lookup_packed and sshash_lookup_ascii on HOST arrays, here page-locked
    ones (the kernels then read the queries and write the ids where they lie: engine.hip host_lookup), wall clock of the call, ids equal
    to the device path's. 16 B per lookup cross the link packed (8 in, 8 out), k + 8 as characters."""
    import torch

    from sshash_amd import _binding as B

    m = min(dq.numel() // W, queries)
    k = d.k()
    lib = B._load()
    want = ids_of_device_path[:m].cpu()
    q_host = dq[: m * W].cpu().pin_memory()
    out_host = torch.zeros(m, dtype=torch.int64).pin_memory()
    res = {}

    def timed(fn, q_ptr):
        r = B._Results()
        r.kmer_id = out_host.data_ptr()
        best = None
        for i in range(4):  # (the first call sizes pools and maps the arrays)
            out_host.fill_(7)
            t0 = time.perf_counter()
            status = fn(d._h, q_ptr, m, 1, B.C.byref(r))
            dt = time.perf_counter() - t0
            B._check(status)
            if i:
                best = dt if best is None else min(best, dt)
        if not torch.equal(out_host, want):
            raise SystemExit("PARITY FAILURE: a host-buffer entry point and the device path disagree")
        return best

    t = timed(lib.sshash_lookup_packed, q_host.data_ptr())
    res["host_packed"] = {"lookups_per_s": round(m / t, 1), "ms": round(t * 1e3, 2), "queries": m, "ids_equal_device_path": True,
                          "link_GBps_both_directions": round(m * (8 * W + 8) / t / 1e9, 1), "caller_arrays": "page-locked, device-mapped"}
    # the same queries as characters (include/kmer.hpp:118: A C T G = 0 1 2 3), k bytes each
    chars = torch.tensor(list(b"ACTG"), dtype=torch.uint8, device=dq.device)
    sh = torch.arange(k, device=dq.device, dtype=torch.int64)
    ascii_host = torch.empty((m, k), dtype=torch.uint8).pin_memory()
    for a in range(0, m, 1 << 22):
        b = min(m, a + (1 << 22))
        words = dq[a * W: b * W].view(-1, W)
        codes = (words[:, (sh >> 5)] >> ((sh & 31) * 2)[None, :]) & 3
        ascii_host[a:b] = chars[codes].cpu()
    del q_host
    t = timed(lib.sshash_lookup_ascii, ascii_host.data_ptr())
    res["host_ascii"] = {"lookups_per_s": round(m / t, 1), "ms": round(t * 1e3, 2), "queries": m, "ids_equal_device_path": True,
                         "link_GBps_both_directions": round(m * (k + 8) / t / 1e9, 1), "caller_arrays": "page-locked, device-mapped"}
    return res


def measure_streaming_from_file(d, index_path, reads_tensor, directory, tag, oracle_sample=100_000, gzip_level=1, log=lambda *a: None):
    """`sshash query` end to end (sshash_streaming_query_from_file: decompress, split, H2D, streaming kernels), wall clock, on a
    FASTQ written from `reads_tensor` -- plain and gzipped --, with the reader alone (tools/reader_rate.cpp: no GPU) and the CPU
    oracle's streaming state machine on the first reads of the same file beside it; counters checked against the oracle."""
    from sshash_amd.synthetic import write_fastq

    k = d.k()
    n, L = reads_tensor.shape
    res = {"reads": int(n), "read_length": int(L), "kmers": int(n) * (L - k + 1)}
    exe = os.path.join(directory, "reader_rate")
    have_exe = subprocess.call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tools", "reader_rate.cpp"),
                                os.path.join(ROOT, "sshash_amd", "csrc", "reads.cpp"), "-lz", "-lpthread", "-o", exe]) == 0
    # plain; gzip (one deflate stream after another: inflated on one thread, like the reference's input); BGZF (bgzip's members: on all cores)
    for flavour, level in (("fastq", None), ("fastq.gz", gzip_level), ("bgzf.fastq.gz", gzip_level)):
        path = os.path.join(directory, f"sshash_amd_{tag}.{flavour}")
        t0 = time.perf_counter()
        size = write_fastq(reads_tensor, path, gzip_level=level, workers=max(1, (os.cpu_count() or 8) // 2), bgzf=flavour.startswith("bgzf"))
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
    if d.k() <= 31:
        res["published_reference"] = {"ns_per_kmer": 89.5, "what": "human k=31 regular, SRR5833294 (91.6 % positive), gzipped FASTQ, one 5.4 GHz core, "
                                      "benchmarks/results-21-01-26/k31/regular-streaming-queries-high-hit.json:3"}
    else:
        res["published_reference"] = {"ns_per_kmer": 190.6, "what": "human k=63 regular, high-hit, 477 818 474 k-mers in 91 062 ms, gzipped FASTQ, one 5.4 GHz "
                                      "core, benchmarks/results-21-01-26/k63/regular-streaming-queries-high-hit.json:3"}
    return res


def traffic_key(args, path=None):
    """name of this run's record in profiles/traffic.json (tools/make_traffic_json.py)"""
    name = args.workload + ("_canonical" if args.canonical else "")
    if path is None and os.environ.get("SSHASH_AMD_SKTABLE", "1") == "0":  # (a replica built without the table: its own records)
        path = "directory" if os.environ.get("SSHASH_AMD_DIRECTORY", "1") != "0" else "mphf"
        if args.streaming:
            return f"{name}_{path}_streaming_p{int(round(args.positive * 100))}"
    if path:
        return f"{name}_{path}"
    return f"{name}_streaming_p{int(round(args.positive * 100))}" if args.streaming else name


def traffic_record(d, n_local: int, args, path=None):
    """HBM bytes per step from the PMC passes of THIS workload (profiles/traffic.json: one record per traffic_key, written by
    tools/make_traffic_json.py from rocprofv3 --pmc runs of this very command); None when no record fits this run's shape (n_local:
    queries, or reads, of this GPU). The provenance (file, commit, counters) is printed with the number."""
    prof = os.path.join(ROOT, "profiles", "traffic.json")
    key = traffic_key(args, path)
    try:
        rec = json.load(open(prof)).get(key)
    except Exception:
        return None, None
    if not rec:
        return None, None
    same = rec.get("canonical") == args.canonical and rec.get("k", 31) == args.k and rec.get("bases") in (None, args.bases, d.num_bases())
    if args.streaming and not path:
        same = same and rec.get("reads") == n_local and rec.get("read_length") == args.read_len
    else:
        same = same and rec.get("queries") == n_local
    if not same:
        return None, None
    return rec.get("hbm_bytes_per_launch"), {"file": "profiles/traffic.json", "record": key, "commit": rec.get("commit"),
                                            "counters": rec.get("counters"), "note": rec.get("unit_note"), "source": rec.get("source")}


def random_line_probe(device=None):
    """What THIS box's memory system sustains when a kernel does nothing but one random 64-byte read per lane group: tools/tlb_probe
    (built by `make -C sshash_amd/csrc`), 2^27 independent reads of a 32 GiB array, every line fetched by four adjacent lanes with one
    load instruction -- the access pattern of the table's bucket fetch. Runs of one command differ by up to 10 % on the headline; this
    puts the line's own box under roofline.random_unit_bound next to the figure the constant above was calibrated with. A side
    measurement in a process of its own, after everything else."""
    tool = os.path.join(ROOT, "tools", "tlb_probe")
    if not os.path.exists(tool):
        return {"error": "tools/tlb_probe is not built (make -C sshash_amd/csrc)"}
    try:
        env = dict(os.environ)
        if device is not None:  # (a rank of an N > 1 run: its own GPU, whatever the launcher's visibility list was)
            visible = [v for v in env.get("HIP_VISIBLE_DEVICES", "").split(",") if v]
            env["HIP_VISIBLE_DEVICES"] = visible[device] if device < len(visible) else str(device)
        # (round 6: a large block sustains one of two rates, 7 % apart, and which one is drawn per allocation -- profiles/r06/alloc_modes_*.txt;
        # the bound is what the memory system CAN sustain: three processes, each with an allocation of its own, the best kept, all reported)
        seen = []
        for _ in range(3):
            p = subprocess.run([tool, "32768", "64", "malloc", "0", "0", str(1 << 27), "5", "coop"], capture_output=True, text=True, timeout=300, env=env)
            lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if p.returncode != 0 or not lines:
                if seen:
                    break
                return {"error": f"tools/tlb_probe: exit code {p.returncode}: {p.stderr[-300:]}"}
            seen.append(json.loads(lines[-1]))
        r = max(seen, key=lambda x: x["Greads_per_s"])
        return {"probe_units_per_s": r["Greads_per_s"] * 1e9, "ms_best": r["ms_best"], "ms_avg": r["ms_avg"],
                "every_allocation_G_per_s": [x["Greads_per_s"] for x in seen],
                "what": "tools/tlb_probe 32768 64 malloc 0 0 134217728 5 coop: 2^27 random 64-byte lines of a 32 GiB array, best of 5 launches, "
                        "best of 3 processes (allocations)"}
    except Exception as e:  # noqa: BLE001 -- a side measurement must not cost the line
        return {"error": f"tools/tlb_probe: {e!r}"}


def run_other_workload(args, extra, env_extra=None):
    """`python bench.py --workload ...` as a child process (its own index, replica and batch; this process has released its GPU
    memory): the child's JSON line, or the reason there is none."""
    record = os.path.join(args.cache_dir, f"sshash_amd_bench_child_{os.getpid()}.json")
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup), "--cache-dir", args.cache_dir,
           "--seed", str(args.seed), "--no-extra-mixes", "--no-other-paths", "--no-file-query", "--no-other-workloads", "--no-line-probe",
           "--full-record", record, "--quiet-record"] + extra
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    t0 = time.time()
    p = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)  # (its log goes where this process's log goes)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if p.returncode != 0 or not lines or not os.path.exists(record):
        return {"error": f"exit code {p.returncode}", "command": " ".join(cmd[1:])}
    line = json.load(open(record))  # the child's FULL record (its stdout line is the compact form of it)
    os.remove(record)
    line["wall_s_of_the_child"] = round(time.time() - t0, 1)
    if env_extra:
        line["environment"] = dict(env_extra)  # a replica configured otherwise than the default (the switches)
    return line


def streaming_roofline(d, args, n_reads_local, W, achieved, avg_kernel_ms, kernel_ms, algorithmic, rep, n_local_kmers, bytes_per_lookup):
    """the roofline object of a streaming line: algorithmic bytes against the HBM peak, the PMC traffic of this very workload
    (profiles/traffic.json) and the 64-byte lines it asks of the memory system against the random-line rate"""
    traffic, provenance = traffic_record(d, n_reads_local, args)
    roof = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": traffic, "traffic_provenance": provenance,
            "kernel": "streaming_run_kernel<W=%d> + stream_pack_kernel (a lane handles the events of its read -- a seed, or the word-wise extension behind a hit --, "
                      "reads packed to 2 bits first; avg_kernel_ms = HIP-event time around one step, on the launch stream)" % W,
            "avg_kernel_ms": round(avg_kernel_ms, 3), "kernel_ms_steps": [round(float(t), 3) for t in kernel_ms],
            "algorithmic_bytes_per_kmer": round(algorithmic / rep["num_kmers"], 3),
            "algorithmic_bytes_rule": "1 B per base + 8 B per distinct 64-bit index word the reference's streaming state machine dereferences per k-mer (the "
                                      "lookups of a seed() its unchanged-minimizer test does not cut short; the strings' next k-mer of an extension), counted "
                                      "by the instrumented oracle on the checked sample of reads",
            #  not a physical bound: the kernel skips reads the reference makes (a miss stands for its neighbours, a run is measured
            # by words), so this fraction can pass 1; the physical fractions are frac_hbm_traffic and random_line_bound
            "frac_is": "reference-algorithm bytes / kernel time / peak -- not an upper bound; physical: frac_hbm_traffic, random_line_bound"}
    if traffic:
        roof["frac_hbm_traffic"] = round(traffic / (avg_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
        roof["hbm_traffic_bytes_per_kmer"] = round(traffic / n_local_kmers, 3)
        # what bounds a kernel of dependent random reads is random LINES, not bytes (tools/tlb_probe): the step's 64-byte requests per second
        # against the rate a kernel doing nothing else sustains -- this box's (attached by the parent run) first, the r02 constant second
        lines = traffic / 64.0
        roof["random_unit_bound"] = {"probe_units_per_s": RANDOM_UNIT_PROBE, "source": "profiles/r02/tlb_probe_128_256_byte_units.jsonl",
                                     "units": "64-byte requests of the step (PMC traffic / 64)", "units_per_s_this_gpu": round(lines / (avg_kernel_ms * 1e-3), 1),
                                     "frac": round(lines / (avg_kernel_ms * 1e-3) / RANDOM_UNIT_PROBE, 4)}
    return roof


def streaming_mode(args, d, index_path, rank, world, local_rank, dev, use_dist, dist, coll_dev, barrier, stats):
    """`--streaming`: BASELINE.json configs[3] -- streaming_query over synthetic reads, read-sharded: ONE set of --reads reads of
    --read-len bases (half -- `--positive` -- spell consecutive k-mers of the dictionary, with 1 % substitutions; the rest random;
    'N' at 1e-3), rank r draws and keeps its share on its device; a step is one sshash_streaming_query_device call over the share
    (the reference's state machine per read, include/streaming_query.hpp:56-197; six counters out). No collective on the data path:
    the counters are summed with one all_reduce after the timed region. Returns the JSON line (rank 0) or None."""
    import torch

    from sshash_amd.synthetic import make_reads_device

    lo, hi = split_batch(args.reads, world, rank)
    n = hi - lo
    L, k = args.read_len, d.k()
    t0 = time.time()
    reads = make_reads_device(d, local_rank, n, L, positive_fraction=args.positive, seed=args.seed + 7919 * rank)
    offsets = torch.arange(n + 1, dtype=torch.int64, device=dev) * L
    report = torch.zeros(6, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    if rank == 0:
        log(f"{n} reads of {L} bases drawn on the device in {time.time() - t0:.1f}s")
    stream = torch.cuda.current_stream()

    def step():
        report.zero_()
        d.streaming_query_device(local_rank, reads.data_ptr(), offsets.data_ptr(), n, report.data_ptr(), stream=stream.cuda_stream, total_bases=n * L)

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
    own_elapsed = time.perf_counter() - t_begin
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_begin
    kernel_ms = [starts[i].elapsed_time(stops[i]) for i in range(args.steps)]
    avg_kernel_ms = float(np.mean(kernel_ms))
    counters = report.clone()
    per_rank = [{"rank": 0, "reads": n, "ms_per_step": round(own_elapsed / args.steps * 1e3, 3), "kernel_ms_per_step": round(avg_kernel_ms, 3),
                 "report": [int(v) for v in counters.cpu().tolist()]}]
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        mine = torch.tensor([float(n), own_elapsed / args.steps * 1e3, avg_kernel_ms] + [float(v) for v in counters.cpu().tolist()], dtype=torch.float64, device=coll_dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [{"rank": r, "reads": int(v[0].item()), "ms_per_step": round(float(v[1].item()), 3), "kernel_ms_per_step": round(float(v[2].item()), 3),
                     "report": [int(x.item()) for x in v[3:]]} for r, v in enumerate(every)]
        total = counters.to(coll_dev)
        dist.all_reduce(total, op=dist.ReduceOp.SUM)  # the one reduction of the read-sharded query: six counters
        counters = total
    if rank != 0:
        return None
    names = ("num_kmers", "num_positive_kmers", "num_negative_kmers", "num_invalid_kmers", "num_searches", "num_extensions")
    rep = dict(zip(names, (int(v) for v in counters.cpu().tolist())))
    # the reference's own consistency rules (src/query.cpp:44-45, include/streaming_query.hpp:113) and the known k-mer count
    assert rep["num_kmers"] == args.reads * (L - k + 1), rep
    assert rep["num_kmers"] == rep["num_positive_kmers"] + rep["num_negative_kmers"] + rep["num_invalid_kmers"], rep
    assert rep["num_positive_kmers"] == rep["num_searches"] + rep["num_extensions"], rep
    assert sum(r["report"][0] for r in per_rank) == rep["num_kmers"]
    # parity: the first reads of rank 0's share through the CPU oracle's restated state machine (oracle = checker only)
    from oracle import oracle as O

    ora = O.OracleIndex(index_path)
    m = min(n, args.stream_oracle_reads)
    sample = [bytes(r) for r in reads[:m].cpu().numpy()]
    t0 = time.perf_counter()
    want = ora.streaming_query(sample)
    t_oracle = time.perf_counter() - t0
    part = torch.zeros(6, dtype=torch.int64, device=dev)
    d.streaming_query_device(local_rank, reads.data_ptr(), offsets.data_ptr(), m, part.data_ptr(), stream=stream.cuda_stream, total_bases=m * L)
    torch.cuda.synchronize()
    got = dict(zip(names, (int(v) for v in part.cpu().tolist())))
    if got != {f: int(v) for f, v in want.items()}:
        raise SystemExit(f"PARITY FAILURE: streaming counters of the first {m} reads: GPU {got} vs oracle {want}")
    # algorithmic bytes (SURVEY 8(d) rule, applied to the streaming query k-mer by k-mer): 1 B per base of the reads + 8 B per distinct
    # 64-bit index word the REFERENCE's state machine dereferences for a k-mer -- the lookups of a seed() that its unchanged-minimizer
    # test does not cut short, the strings' next k-mer of an extension --, counted by the instrumented oracle on the sample it has just
    # checked (the first `m` reads of rank 0's share) and scaled by k-mers. (Until round 5 every negative was priced as a full lookup,
    # which the reference does not do either: a line that skips what the reference skips then sat above the roofline.)
    W = d.words_per_kmer()
    bytes_per_kmer = ora.streaming_count_bytes(sample) / max(1, want["num_kmers"])
    algorithmic = bytes_per_kmer * rep["num_kmers"]
    bytes_per_lookup = None
    n_local_kmers = per_rank[0]["report"][0]
    achieved = algorithmic * (n_local_kmers / rep["num_kmers"]) / (avg_kernel_ms * 1e-3) / 1e9
    total_kmers = rep["num_kmers"] * args.steps
    return {
        "metric": "streaming_query k-mers/sec (synthetic reads resident in HBM, six counters; BASELINE.json configs[3])",
        "value": round(total_kmers / elapsed, 1), "unit": "k-mers/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"{WORKLOADS[args.workload][3]}, k={k} m={d.m()} {'canonical' if d.canonical() else 'regular'}, {d.num_kmers()} k-mers; "
                               f"streaming_query over ONE set of {args.reads} reads x {L} bases ({args.positive:.0%} spell k-mers of the dictionary with 1 % "
                               f"substitutions, the rest random; N at 1e-3), read-sharded over {world} GPU(s), index replicated",
                   "workload_short": f"{args.workload.upper()} stand-in, k={k} m={d.m()} {'canonical' if d.canonical() else 'regular'}, {d.num_kmers()} k-mers; streaming_query "
                                     f"over ONE set of {args.reads} reads x {L} bases ({args.positive:.0%} from the dictionary with 1 % substitutions, rest random, N at "
                                     f"1e-3), read-sharded over {world} GPU(s), index replicated",
                   "reads": args.reads, "read_length": L, "reads_per_gpu": n, "k": k, "m": d.m(), "canonical": d.canonical(), "num_kmers": d.num_kmers(),
                   "device_index_bytes": d.device_bytes(local_rank), "report": rep,
                   "positive_fraction_of_kmers": round(rep["num_positive_kmers"] / rep["num_kmers"], 4),
                   "extensions_per_search": round(rep["num_extensions"] / max(1, rep["num_searches"]), 2),
                   "counters_equal_oracle_on_reads": m, "parity": PARITY_NOTE, "num_bases": d.num_bases(), "device_stats": stats},
        "per_rank": per_rank,
        "traffic_key": traffic_key(args),
        "roofline": streaming_roofline(d, args, n, W, achieved, avg_kernel_ms, kernel_ms, algorithmic, rep, n_local_kmers, bytes_per_lookup),
        "cpu_baseline": {"value": round(want["num_kmers"] / t_oracle, 1), "unit": "k-mers/s", "cores": 1, "kind": "port",
                         "sample": f"the first {m} reads of the same set through the oracle's streaming state machine on one thread (the reference's query "
                                   f"tool is single-threaded): {t_oracle / max(1, want['num_kmers']) * 1e9:.1f} ns per k-mer",
                         "published_reference_ns_per_kmer": "89.5 (k=31) / 190.6 (k=63): human, high-hit, gzipped FASTQ, one 5.4 GHz core; "
                                                            "benchmarks/results-21-01-26/k{31,63}/regular-streaming-queries-high-hit.json:3"},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c3")
    ap.add_argument("--bases", type=int, default=None, help="bases in the synthetic SPSS (default: the workload's)")
    ap.add_argument("--queries", type=int, default=None, help="queries in the batch, ALL GPUs together (default: the workload's)")
    ap.add_argument("--k", type=int, default=None, help="default: the recipe's (31; 63 for --workload c4)")
    ap.add_argument("--m", type=int, default=None, help="default: the recipe's (21; 25 for --workload c4)")
    ap.add_argument("--recipe", default=None, help="sshash_amd/recipes/<name>.json (default: the workload's)")
    ap.add_argument("--repeat-scale", type=float, default=1.0, help="multiply the amount of every repeat family of the recipe")
    ap.add_argument("--no-file-query", action="store_true", help="skip the end-to-end FASTQ query (side measurement, outside the timed region)")
    ap.add_argument("--file-reads", type=int, default=2_000_000, help="reads of the end-to-end FASTQ query (tools/bench_streaming_file.py runs 10^8)")
    ap.add_argument("--no-other-paths", action="store_true", help="skip the table-less paths (side measurement, outside the timed region)")
    ap.add_argument("--canonical", action="store_true")
    ap.add_argument("--positive", type=float, default=0.5, help="fraction of positive queries in the batch")
    ap.add_argument("--negatives", choices=["random", "mutated"], default="random",
                    help="negative queries: uniform random k-mers (the reference's protocol) or indexed k-mers with one substitution")
    ap.add_argument("--seed", type=int, default=0x5555AAAA)
    ap.add_argument("--cache-dir", default=os.environ.get("SSHASH_BENCH_CACHE", "/tmp"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-mixes", action="store_true", help="skip the untimed-region side measurements (other query mixes)")
    ap.add_argument("--cpu-sample", type=int, default=2_000_000, help="queries per CPU-baseline pass")
    ap.add_argument("--sharded", choices=["table", "minimizer"], default=None,
                    help="NOT the headline mode: partition the dictionary over the GPUs instead of replicating it (BASELINE.json "
                         "configs[4]) -- 'table': the super-k-mer table by key, 'minimizer': the minimizer-side structures -- and "
                         "route every query to its owner with an all-to-all over RCCL (sshash_sharded_lookup_device)")
    ap.add_argument("--streaming", action="store_true",
                    help="measure the streaming query (BASELINE.json configs[3]) instead of the point lookup: --reads reads of --read-len bases, read-sharded")
    ap.add_argument("--reads", type=int, default=100_000_000, help="--streaming: reads in the set, ALL GPUs together")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--stream-oracle-reads", type=int, default=50_000, help="--streaming: reads checked against (and timed on) the CPU oracle")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="default run (C3, one GPU): do not append the C2 and C4 lines (`other_workloads`: child runs of this script)")
    ap.add_argument("--other-streaming-reads", type=int, default=20_000_000, help="reads of the C4 streaming line inside `other_workloads`")
    ap.add_argument("--no-line-probe", action="store_true", help="skip the random-line probe of this box (tools/tlb_probe; side measurement after everything else)")
    ap.add_argument("--full-record", default=os.path.join(ROOT, "bench_full.json"),
                    help="where rank 0 writes the FULL record (children, histograms, per-step times, the rules in prose); stdout carries the compact line")
    ap.add_argument("--quiet-record", action="store_true", help="do not echo the full record on stderr")
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()
    bases, recipe, queries, what = WORKLOADS[args.workload]
    if args.bases is None:
        args.bases = bases
    if args.recipe is None:
        args.recipe = recipe
    if args.k is None or args.m is None:
        from sshash_amd.repeats import load_recipe

        r = load_recipe(args.recipe)
        args.k = r["k"] if args.k is None else args.k
        args.m = r["m"] if args.m is None else args.m
    if args.k > 31 and args.recipe.endswith("_k31") and os.path.exists(os.path.join(ROOT, "sshash_amd", "recipes", "human_k63.json")):
        args.recipe = "human_k63"  # (`--k 63` on a k = 31 workload: the k = 63 recipe at that workload's size)
    if args.queries is None:
        args.queries = queries

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:]))

    # stdout carries the ONE JSON line and nothing else: whatever the libraries write to file descriptor 1 meanwhile
    # (RCCL prints its version banner and its warnings there) is sent to stderr until the line is due
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test scaffolding (tests/test_gpu_bench_harness.py): on a one-GPU box the N > 1 harness is exercised with every rank
    # on the same device and the coordination over gloo; never set in a measurement
    one_device = os.environ.get("SSHASH_BENCH_TEST_ALL_RANKS_ON_DEVICE")
    if one_device is not None:
        local_rank = int(one_device)
    backend = "gloo" if one_device is not None else "nccl"
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
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
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")  # where the tensors of the (few) collectives live

    def barrier():
        if use_dist:
            dist.barrier()

    from sshash_amd.synthetic import draw_queries_device

    d, index_path = get_index(args, rank, world, barrier)
    t0 = time.time()
    sharded = None
    if args.sharded:
        if not use_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            dist.init_process_group(backend="nccl", device_id=dev, rank=0, world_size=1)
            use_dist = True
        from sshash_amd.sharded import ShardedDictionary

        if args.sharded == "table":
            sharded = ShardedDictionary(d, local_rank, by="table")
        else:
            # rank r keeps the buckets of its own minimizers only: rebuilt from the cached dictionary's strings
            import sshash_amd

            words, endpoints = make_standin(args)
            shard = sshash_amd.Dictionary.build_from_packed(words, endpoints, k=args.k, m=args.m, canonical=args.canonical, num_threads=0,
                                                            num_shards=world, shard_id=rank)
            del words
            sharded = ShardedDictionary(shard, local_rank, by="minimizer")
            d.to_device(local_rank)  # (the complete dictionary: used to draw the queries)
    else:
        d.to_device(local_rank)
    upload_window = (t0, time.time())  # wall clock of this node: the ranks' uploads must overlap, not queue (per_rank)
    stats = d.device_stats(local_rank)
    if rank == 0:
        log(f"replica in HBM: {d.device_bytes(local_rank) / 1e6:.0f} MB (upload {upload_window[1] - t0:.1f}s); {stats}")

    if args.streaming:
        result = streaming_mode(args, d, index_path, rank, world, local_rank, dev, use_dist, dist, coll_dev, barrier, stats)
        barrier()
        if use_dist:
            dist.destroy_process_group()
        emit(json_fd, rank, result, args)
        return

    # this rank's share of the batch
    lo, hi = split_batch(args.queries, world, rank)
    n = hi - lo
    W = d.words_per_kmer()
    t0 = time.time()
    dq = draw_queries_device(d, local_rank, n, args.positive, seed=args.seed + 7919 * rank, negatives=args.negatives)
    torch.cuda.synchronize()
    if rank == 0:
        log(f"{n} queries drawn on the device in {time.time() - t0:.1f}s")
    out = torch.empty(n, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream()

    def step(q=dq, o=out, count=n):
        if sharded is not None:
            o[:count] = sharded.lookup_device(q[: count * W])
        else:
            d.lookup_device(local_rank, q.data_ptr(), count, o.data_ptr(), check_reverse_complement=True, stream=stream.cuda_stream)

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
    own_elapsed = time.perf_counter() - t_begin
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_begin
    kernel_ms = [starts[i].elapsed_time(stops[i]) for i in range(args.steps)]
    avg_kernel_ms = float(np.mean(kernel_ms))
    per_rank = [{"rank": 0, "queries": n, "ms_per_step": round(own_elapsed / args.steps * 1e3, 3), "kernel_ms_per_step": round(avg_kernel_ms, 3),
                 "upload_s": round(upload_window[1] - upload_window[0], 2)}]
    kernel_ms_steps = [round(float(t), 3) for t in kernel_ms]  # rank 0's steps one by one: sustained load drifts (clocks), see HISTORY.md
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # every rank's own GPU under the random-line probe (the boxes -- and the GPUs of one node -- differ by several per cent): after
        # the timed region, all ranks at once, each on its own device
        probe = {} if (args.no_line_probe or world == 1) else random_line_probe(device=local_rank)
        mine = torch.tensor([float(n), own_elapsed / args.steps * 1e3, avg_kernel_ms, upload_window[0], upload_window[1],
                             float(probe.get("probe_units_per_s", 0.0))], dtype=torch.float64, device=coll_dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        first = min(float(v[3].item()) for v in every)
        per_rank = [{"rank": r, "queries": int(v[0].item()), "ms_per_step": round(float(v[1].item()), 3),
                     "kernel_ms_per_step": round(float(v[2].item()), 3), "upload_s": round(float((v[4] - v[3]).item()), 2),
                     "upload_window_s": [round(float(v[3].item()) - first, 2), round(float(v[4].item()) - first, 2)],
                     "random_line_probe_units_per_s": float(v[5].item()) or None} for r, v in enumerate(every)]

    # ---- parity spot check + algorithmic bytes (oracle = checker only) ---------------------------
    result = None
    if rank == 0:
        from oracle import oracle as O

        ora = O.OracleIndex(index_path)
        sample = min(200_000, n)
        head = dq[: max(sample, min(n, args.cpu_sample * 64)) * W].cpu().numpy().view(np.uint64)
        got = out[:sample].cpu().numpy().view(np.uint64)
        want = ora.lookup_ids(head[: sample * W], num_threads=effective_cores())
        if not (got == want).all():
            bad = np.nonzero(got != want)[0]
            raise SystemExit(f"PARITY FAILURE: GPU ids differ from the CPU oracle for {bad.size} of {sample} sampled queries; first: "
                             + ", ".join(f"#{int(i)} kmer={int(head[i * W]):#x} got={int(got[i])} want={int(want[i])}" for i in bad[:6]))
        found = float((out != -1).float().mean().item())
        bytes_per_lookup = ora.count_bytes(head[: min(sample, 100_000) * W]) / min(sample, 100_000)
        achieved = bytes_per_lookup * n / (avg_kernel_ms * 1e-3) / 1e9
        pieces = -(-n // (1 << 27))  # launch sequences per step (engine.hip: at most 2^27 queries per sequence, equal pieces)
        traffic, provenance = traffic_record(d, n, args)
        roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_provenance": provenance,
                    "kernel": "fast_lookup_kernel<W=%d,canonical=%d,ids,%s> (%s) + deferred_lookup_kernel (%d launch "
                              "sequence(s) of equal size per step; avg_kernel_ms = HIP-event time around one step, on the launch stream)"
                              % (W, int(d.canonical()), "super-k-mer table" if stats["sk_slots"] else "directory / MPHF",
                                 ("every probe finished inside the first pass" if W == 1 else "+ resume_lookup_kernel") if stats["sk_slots"]
                                 else "+ scan_lookup_kernel", pieces),
                    "launches_per_step": pieces,
                    "algorithmic_bytes_per_lookup": round(bytes_per_lookup, 2),
                    "algorithmic_bytes_rule": "SURVEY 8(d): 8 B per distinct 64-bit index word the REFERENCE algorithm dereferences "
                                              "+ query in + id out, counted by the instrumented oracle on this batch",
                    "avg_kernel_ms": round(avg_kernel_ms, 3), "kernel_ms_steps": kernel_ms_steps,
                    # what bounds a structure of one random bucket per query on this chip is random UNITS, not bytes: 43.8 G
                    # random 64-byte units/s, 39.9 G 128-byte ones (tools/tlb_probe; HISTORY.md)
                    "random_unit_bound": {"probe_units_per_s": RANDOM_UNIT_PROBE, "source": "profiles/r02/tlb_probe_128_256_byte_units.jsonl",
                                          "lookups_per_s_this_gpu": round(n / (avg_kernel_ms * 1e-3), 1), "units_per_s_this_gpu": round(n / (avg_kernel_ms * 1e-3), 1),
                                          "frac": round(n / (avg_kernel_ms * 1e-3) / RANDOM_UNIT_PROBE, 4)}}
        if traffic:
            # the kernels' own bytes: what the PMC passes of this very workload saw moving between L2 and HBM per step
            # (64 B per bucket line -- both slots of a line are compared --, the query and id streams, the pass queues)
            roofline["frac_hbm_traffic"] = round(traffic / (avg_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
            roofline["hbm_traffic_bytes_per_lookup"] = round(traffic / n, 2)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cores = effective_cores()
            m = min(args.cpu_sample, n)
            t0 = time.perf_counter()
            ora.lookup_ids(head[: m * W], num_threads=1)
            t1 = time.perf_counter() - t0
            reps = max(1, min(64, int(20.0 / max(1e-3, t1 / cores * 4))))
            big = min(head.size // W, m * reps)
            t0 = time.perf_counter()
            ids_all = ora.lookup_ids(head[: big * W], num_threads=cores)
            ta = time.perf_counter() - t0
            assert (ids_all == out[:big].cpu().numpy().view(np.uint64)).all()
            cpu = {"value": round(big / ta, 1), "unit": "lookups/s", "cores": cores, "kind": "port",
                   "sample": f"{big} queries of the same batch on {cores} threads (= usable CPUs: affinity {os.cpu_count()}, "
                             f"cgroup quota applied; contiguous chunks); "
                             f"1 thread: {m / t1:.0f} lookups/s = {t1 / m * 1e9:.0f} ns/lookup over {m} queries",
                   "single_thread_value": round(m / t1, 1),
                   "published_reference_ns_per_lookup": "738-768 ns (real S. enterica index, one 5.4 GHz core; BASELINE.md)"}
        extra = None
        if world == 1 and not args.no_extra_mixes:
            # side measurements OUTSIDE the timed region: other query mixes on 10^8-query batches, ids of a sample checked
            extra = {}
            m = min(n, 100_000_000)
            for name, pf, neg in (("positive100", 1.0, "random"), ("negative100_random", 0.0, "random"),
                                  ("mix50_mutated_negatives", 0.5, "mutated")):
                q2 = draw_queries_device(d, local_rank, m, pf, seed=args.seed + 101, negatives=neg)
                o2 = out[:m]
                step(q2, o2, m)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(3):
                    step(q2, o2, m)
                e1.record(stream)
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 3
                s2 = min(m, 50_000)
                w2 = ora.lookup_ids(q2[: s2 * W].cpu().numpy().view(np.uint64), num_threads=effective_cores())
                if not (o2[:s2].cpu().numpy().view(np.uint64) == w2).all():
                    raise SystemExit(f"PARITY FAILURE in side measurement {name}")
                extra[name] = {"lookups_per_s": round(m / ms * 1e3, 1), "ms": round(ms, 3), "queries": m,
                               "fraction_found": round(float((o2 != -1).float().mean().item()), 4)}
                del q2
        other_paths = None
        if world == 1 and sharded is None and not args.no_other_paths and stats["sk_slots"]:
            other_paths = measure_other_paths(args, index_path, local_rank, d, dq, W, bytes_per_lookup)
        from_file = None
        if world == 1 and sharded is None and not args.no_file_query:
            from sshash_amd.synthetic import make_reads_device

            reads = make_reads_device(d, local_rank, args.file_reads, 150, positive_fraction=0.9, seed=args.seed + 5).cpu()
            from_file = measure_streaming_from_file(d, index_path, reads, args.cache_dir, f"bench{os.getpid()}", oracle_sample=50_000, log=log)
            from_file["workload"] = (f"{args.file_reads} reads x 150 bp, 90 % drawn from the dictionary with 1 % substitutions, N at 1e-3 per base; "
                                     "tools/bench_streaming_file.py runs 10^8 reads (profiles/r03/)")
            del reads
        index_statistics = table_histogram = None
        from sshash_amd.repeats import load_recipe

        if int(load_recipe(args.recipe)["k"]) == d.k():
            from sshash_amd.repeats import statistics_vs_target

            index_statistics = statistics_vs_target(d.bucket_stats(), args.recipe)
            table_histogram = d.device_table_histogram(local_rank)
            heavy = sum(v for kk, v in table_histogram["super_kmers_by_occurrences_of_their_key"].items() if kk not in ("1", "2", "3", "4"))
            table_histogram["super_kmers_under_heavy_keys_fraction"] = round(heavy / max(1, table_histogram["super_kmers"]), 5)
            table_histogram["kmers_under_heavy_keys"] = stats["sk_heavy_kmers"]
            table_histogram["kmers_under_heavy_keys_fraction"] = round(stats["sk_heavy_kmers"] / d.num_kmers(), 5)
        total = args.queries * args.steps
        result = {
            "metric": "k-mer Lookups/sec (batched random queries, bit-exact ids)",
            "value": round(total / elapsed, 1),
            "unit": "lookups/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": f"{what}, k={d.k()} m={d.m()} {'canonical' if d.canonical() else 'regular'}, "
                                   f"{d.num_kmers()} k-mers / {d.num_strings()} strings / {d.num_bases()} bases, index replicated in "
                                   f"every GPU's HBM; ONE batch of {args.queries} packed queries per step "
                                   f"({args.positive:.0%} positive, half of them reverse-complemented; negatives: {args.negatives}) "
                                   f"split over {world} GPU(s)",
                       "workload_short": f"{args.workload.upper()} stand-in (recipe {args.recipe}: bucket statistics fitted to the reference's build log), k={d.k()} m={d.m()} "
                                         f"{'canonical' if d.canonical() else 'regular'}, {d.num_kmers()} k-mers, index replicated per GPU; ONE batch of "
                                         f"{args.queries} packed queries per step ({args.positive:.0%} positive, half reverse-complemented; negatives {args.negatives}) "
                                         f"split over {world} GPU(s)",
                       "ids_equal_oracle_on_queries": sample, "parity": PARITY_NOTE, "num_bases": d.num_bases(),
                       "queries_per_step": args.queries, "queries_per_gpu": n, "num_kmers": d.num_kmers(), "k": d.k(), "m": d.m(),
                       "canonical": d.canonical(), "index_replicated_per_gpu": sharded is None, "sharded": args.sharded,
                       # what carries the exchange of a routed lookup: torch.distributed's all_to_all_single over the group's backend -- nccl
                       # (= RCCL over xGMI) whenever every rank has a GPU of its own; gloo only under the tests' one-device scaffolding
                       "exchange": (f"all_to_all_single over {backend}" + (" (TEST scaffolding: every rank on one device)" if one_device is not None else "")) if sharded is not None else None,
                       "positive_fraction_found": round(found, 4), "device_index_bytes": d.device_bytes(local_rank),
                       "device_bytes_per_kmer": round(d.device_bytes(local_rank) / d.num_kmers(), 2),
                       "device_stats": stats, "recipe": args.recipe, "repeat_scale": args.repeat_scale,
                       "index_statistics": index_statistics, "table_histogram": table_histogram},
            "traffic_key": traffic_key(args),
            "per_rank": per_rank,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "cpu_baseline_note": None if cpu is not None or args.no_cpu_baseline else "the CPU path is timed by the N = 1 run only (rank 0, this box's host cores): see that run's line",
            "other_mixes": extra,
            "other_paths": other_paths,
            "streaming_from_file": from_file,
            "other_workloads": None,
        }
        other_workloads = None  # (last: this process gives its GPU memory back first)
        # test scaffolding (tests/test_gpu_bench_harness.py): "bases,queries,reads" shrinks the children and lets a reduced parent have them
        reduced = os.environ.get("SSHASH_BENCH_TEST_OTHER_WORKLOADS")
        if world == 1 and sharded is None and args.workload == "c3" and not args.no_other_workloads and (args.bases == WORKLOADS["c3"][0] or reduced):
            # BASELINE.json's other single-GPU configurations, each a run of this script of its own (index, replica, batch, oracle check,
            # roofline, cpu_baseline): C2 (configs[1]) and C4 (configs[3]: the k = 63 dictionary -- point lookups and its streaming query)
            d.close()
            torch.cuda.empty_cache()
            other_workloads = {}
            high_hit = ["--workload", "c3", "--streaming", "--positive", "0.95", "--reads", str(args.other_streaming_reads)]
            for name, extra, env_extra in (("c2", ["--workload", "c2"], None), ("c4", ["--workload", "c4"], None),
                                           ("c4_streaming", ["--workload", "c4", "--streaming", "--reads", str(args.other_streaming_reads)], None),
                                           # the streaming query at k = 31 in the regime the reference publishes (high-hit: 95 % of the reads spell
                                           # k-mers of the dictionary), on the headline's own dictionary
                                           ("c3_streaming_high_hit", high_hit, None)):
                if reduced:
                    b_, q_, r_ = reduced.split(",")
                    extra = extra + ["--bases", b_, "--cpu-sample", "100000"] + (["--reads", r_, "--stream-oracle-reads", "5000"] if "--streaming" in extra else ["--queries", q_])
                log(f"other workload {name} ...")
                other_workloads[name] = run_other_workload(args, extra, env_extra)
                log(f"other workload {name}: {other_workloads[name].get('value')} {other_workloads[name].get('unit')}")
        result["other_workloads"] = other_workloads
        if world == 1 and sharded is None and not args.no_line_probe and isinstance(result.get("roofline"), dict) and "random_unit_bound" in result["roofline"]:
            log("random-line probe of this box ...")
            probe = random_line_probe()
            for line in [result] + [w for w in (other_workloads or {}).values() if isinstance(w, dict)]:
                bound = (line.get("roofline") or {}).get("random_unit_bound")
                if bound is not None:
                    bound["this_box"] = dict(probe)
                    if "probe_units_per_s" in probe:
                        bound["this_box"]["frac"] = round(bound["units_per_s_this_gpu"] / probe["probe_units_per_s"], 4)
            log(f"random-line probe: {probe}")
    barrier()
    if use_dist:
        dist.destroy_process_group()
    emit(json_fd, rank, result, args)


STDOUT_LINE_LIMIT = 8192  # bytes; the driver parses the ONE stdout line, and a line that outgrows its buffer is a lost measurement


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def compact_roofline(roof):
    """The roofline object of the stdout line: the contract's six keys plus the few numbers they were computed from. The prose
    (rules, kernel descriptions, provenance) and the per-step times stay in the full record."""
    if not isinstance(roof, dict):
        return roof
    out = _pick(roof, ("bound", "achieved", "peak", "unit", "frac", "traffic", "avg_kernel_ms", "launches_per_step", "algorithmic_bytes_per_lookup",
                       "algorithmic_bytes_per_kmer", "frac_hbm_traffic", "hbm_traffic_bytes_per_lookup", "hbm_traffic_bytes_per_kmer"))
    out["kernel"] = str(roof.get("kernel", "")).split(" (")[0][:96]
    prov = roof.get("traffic_provenance")
    if isinstance(prov, dict):  # `traffic` is a look-up in a tracked file of builder-run PMC passes, not counters of THIS run: say so beside the number
        out["traffic_source"] = f"{prov.get('file')}[{prov.get('record')}]@{prov.get('commit')}"
    if roof.get("frac_is"):
        out["frac_is"] = roof["frac_is"]
    bound = roof.get("random_unit_bound")
    if isinstance(bound, dict):
        box = bound.get("this_box") or {}
        out["random_line_bound"] = {"units_per_s_this_box": box.get("probe_units_per_s"), "frac_this_box": box.get("frac"),
                                    "units_per_s_r02_constant": bound.get("probe_units_per_s"), "frac_r02_constant": bound.get("frac")}
    return out


def compact_cpu(cpu):
    if not isinstance(cpu, dict):
        return cpu
    out = _pick(cpu, ("value", "unit", "cores", "kind", "single_thread_value", "measured_in"))
    out["sample"] = str(cpu.get("sample", ""))[:160]
    return out


def compact_line(full, record_path):
    """What goes to stdout: the task contract's keys, `roofline`, `cpu_baseline`, the per-rank times, and one short entry per side
    measurement -- a few hundred bytes each. Everything else (index statistics, table histogram, device_stats, per-step kernel
    times, the complete child lines, the rules in prose) is in the full record at `record_path`."""
    if full is None:
        return None
    line = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "traffic_key"))
    cfg = full.get("config", {})
    line["config"] = _pick(cfg, ("queries_per_step", "queries_per_gpu", "reads", "read_length", "reads_per_gpu", "num_kmers", "k", "m", "canonical",
                                 "num_bases", "index_replicated_per_gpu", "sharded", "exchange", "positive_fraction_found", "positive_fraction_of_kmers", "extensions_per_search",
                                 "counters_equal_oracle_on_reads", "ids_equal_oracle_on_queries", "parity", "device_index_bytes", "device_bytes_per_kmer", "recipe",
                                 "report"))
    line["config"] = {"workload": str(cfg.get("workload_short") or cfg.get("workload", ""))[:400], **line["config"]}
    line["per_rank"] = [_pick(r, ("rank", "queries", "reads", "ms_per_step", "kernel_ms_per_step", "random_line_probe_units_per_s")) for r in full.get("per_rank") or []]
    line["roofline"] = compact_roofline(full.get("roofline"))
    line["cpu_baseline"] = compact_cpu(full.get("cpu_baseline"))
    if full.get("cpu_baseline_note"):
        line["cpu_baseline_note"] = full["cpu_baseline_note"]
    if full.get("other_mixes"):
        line["other_mixes"] = {k: v.get("lookups_per_s") for k, v in full["other_mixes"].items()}
    if full.get("other_paths"):
        line["other_paths"] = {k: _pick(v, ("lookups_per_s", "roofline_frac", "ids_equal_table_path", "hbm_traffic_bytes_per_lookup", "ids_equal_device_path", "link_GBps_both_directions")) for k, v in full["other_paths"].items()}
    f = full.get("streaming_from_file")
    if f:
        line["streaming_from_file"] = {fl: {"ns_per_kmer": f[fl]["ns_per_kmer"]} for fl in ("fastq", "fastq.gz", "bgzf.fastq.gz") if fl in f}
        line["streaming_from_file"]["counters_equal_oracle_on_sample"] = f.get("counters_equal_oracle_on_sample")
    if full.get("other_workloads"):
        line["other_workloads"] = {}
        for name, w in full["other_workloads"].items():
            if not isinstance(w, dict) or "error" in w:
                line["other_workloads"][name] = w
                continue
            roof, cpu, wcfg = w.get("roofline") or {}, w.get("cpu_baseline") or {}, w.get("config") or {}
            line["other_workloads"][name] = {
                "value": w.get("value"), "unit": w.get("unit"), "ms_per_step": w.get("ms_per_step"), "steps": w.get("steps"),
                "roofline_frac": roof.get("frac"), "frac_hbm_traffic": roof.get("frac_hbm_traffic"), "traffic": roof.get("traffic"),
                "cpu_baseline_value": cpu.get("value"), "cpu_cores": cpu.get("cores"),
                # every child compares its own results with the oracle before it prints (a mismatch is exit code 1 = "error" here)
                "parity": "ids equal oracle" if "ids_equal_oracle_on_queries" in wcfg else ("counters equal oracle" if "counters_equal_oracle_on_reads" in wcfg else None)}
    line["full_record"] = record_path
    return line


def emit(json_fd, rank, result, args):
    """stdout back in place; rank 0 writes the full record to --full-record (and to stderr) and prints the ONE compact JSON line."""
    sys.stdout.flush()
    try:
        import ctypes

        ctypes.CDLL(None).fflush(None)  # the C library's own buffer of stdout (RCCL's banner sits there until exit)
    except Exception:
        pass
    os.dup2(json_fd, 1)
    if rank != 0:
        return
    path = args.full_record
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as fh:
            json.dump(result, fh)
            fh.write("\n")
    except OSError as e:  # a read-only checkout must not cost the line
        log(f"full record not written to {path}: {e!r}")
        path = None
    line = compact_line(result, path and os.path.relpath(path, ROOT) if path and os.path.abspath(path).startswith(ROOT) else path)
    text = json.dumps(line, separators=(",", ":"))
    if len(text) >= STDOUT_LINE_LIMIT:  # never silently: shed the side measurements, keep the contract
        log(f"stdout line of {len(text)} bytes exceeds {STDOUT_LINE_LIMIT}: side measurements dropped from it (they are in the full record)")
        for key in ("other_workloads", "streaming_from_file", "other_paths", "other_mixes", "per_rank"):
            line.pop(key, None)
            text = json.dumps(line, separators=(",", ":"))
            if len(text) < STDOUT_LINE_LIMIT:
                break
    if not args.quiet_record:
        print("[bench] full record: " + json.dumps(result), file=sys.stderr, flush=True)
    if result is not None:
        roof = result.get("roofline") or {}
        log(f"headline {args.workload}{' streaming' if args.streaming else ''}: {result['value']:.4g} {result['unit']}, {result['ms_per_step']} ms/step, "
            f"frac {roof.get('frac')}, n_gpus {result['n_gpus']}; full record: {path}")
    print(text, flush=True)


if __name__ == "__main__":
    main()
