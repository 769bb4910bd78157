"""CPU, world_size 2, gloo: the multi-process harness of bench.py (N > 1 path).

What N > 1 adds to the single-GPU path is coordination only (the lookup itself shards with no
collective): rank 0 builds and caches the index, every rank loads the same file after a barrier, each
rank draws its OWN query batch, elapsed time is MAX-reduced, totals are summed. That logic is exercised
here with the CPU oracle standing in for the device (no GPU in this container)."""
from __future__ import annotations

import os
import subprocess
import sys
import textwrap

from conftest import ROOT

WORKER = textwrap.dedent(
    """
    import argparse, os, sys, time
    import numpy as np
    sys.path.insert(0, sys.argv[1])
    import torch, torch.distributed as dist
    import bench
    from sshash_amd.synthetic import draw_queries
    from oracle import oracle as O

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo")
    args = argparse.Namespace(bases=300_000, k=31, m=15, canonical=False, seed=77, cache_dir=sys.argv[2], verbose=False, recipe="human_k31",
                              repeat_scale=1.0)
    d, path = bench.get_index(args, rank, world, dist.barrier)
    # every rank sees the same dictionary
    sig = torch.tensor([d.num_kmers(), d.num_strings(), int(d.access_packed(np.arange(0, d.num_kmers(), 101)).sum() % (1 << 62))], dtype=torch.int64)
    gathered = [torch.zeros_like(sig) for _ in range(world)]
    dist.all_gather(gathered, sig)
    assert all(torch.equal(g, gathered[0]) for g in gathered), "ranks loaded different dictionaries"
    # each rank draws its own batch (different seed), looks it up, totals are reduced
    q = draw_queries(d, 20000, 0.5, seed=args.seed + 7919 * rank)
    first = torch.tensor([int(q[0] >> 1)], dtype=torch.int64)
    firsts = [torch.zeros_like(first) for _ in range(world)]
    dist.all_gather(firsts, first)
    assert len({int(f) for f in firsts}) == world, "ranks drew the same batch"
    t0 = time.perf_counter()
    ids = O.OracleIndex(path).lookup_ids(q)
    elapsed = torch.tensor([time.perf_counter() - t0 + 0.01 * rank], dtype=torch.float64)
    mine = float(elapsed)
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    assert float(elapsed) >= mine
    found = torch.tensor([int((ids != np.uint64(0xFFFFFFFFFFFFFFFF)).sum())], dtype=torch.int64)
    dist.all_reduce(found)
    if rank == 0:
        assert int(found) == 10000 * world, int(found)
        print("OK", int(found), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    """
)


def test_two_rank_harness_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    cache = tmp_path / "cache"
    cache.mkdir()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = []
    for rank in range(2):
        e = dict(env, RANK=str(rank), LOCAL_RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(cache)], env=e, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    assert "OK 20000" in outs[0]
    assert len([f for f in os.listdir(cache) if f.endswith(".sshash")]) == 1  # built once, by rank 0


def test_effective_cores_is_positive():
    sys.path.insert(0, ROOT)
    import bench

    assert bench.effective_cores() >= 1


SPAWNED = textwrap.dedent(
    """
    import os, sys
    import torch, torch.distributed as dist
    dist.init_process_group(backend="gloo")   # RANK / WORLD_SIZE / MASTER_* come from torch.distributed.run
    t = torch.tensor([dist.get_rank() + 1])
    dist.all_reduce(t)
    assert os.environ["MASTER_ADDR"] == "127.0.0.1"
    open(os.path.join(sys.argv[1], f"rank{dist.get_rank()}.{int(t)}.{' '.join(sys.argv[2:])}"), "w").close()
    dist.destroy_process_group()
    """
)


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` outside torch.distributed.run spawns the N ranks itself (bench.spawn_ranks): same
    launcher, a stand-in script (no GPU here)."""
    sys.path.insert(0, ROOT)
    import bench

    script = tmp_path / "spawned.py"
    script.write_text(SPAWNED)
    env_before = {k: os.environ.get(k) for k in ("RANK", "WORLD_SIZE")}
    assert env_before == {"RANK": None, "WORLD_SIZE": None}
    rc = bench.spawn_ranks(2, str(script), [str(tmp_path), "--gpus", "2"])
    assert rc == 0
    made = sorted(f for f in os.listdir(tmp_path) if f.startswith("rank"))
    assert made == ["rank0.3.--gpus 2", "rank1.3.--gpus 2"]


def test_strong_scaling_shares_cover_the_batch():
    sys.path.insert(0, ROOT)
    import bench

    for total, world in ((1_000_000_000, 8), (100_000_007, 3), (5, 8)):
        shares = [bench.split_batch(total, world, r) for r in range(world)]
        assert shares[0][0] == 0 and shares[-1][1] == total
        assert all(shares[r][1] == shares[r + 1][0] for r in range(world - 1))
        assert max(hi - lo for lo, hi in shares) - min(hi - lo for lo, hi in shares) <= 1


STREAMING_WORKER = textwrap.dedent(
    """
    import gzip, os, sys
    import numpy as np
    sys.path.insert(0, sys.argv[1])
    import torch, torch.distributed as dist
    import bench, sshash_amd
    from oracle import oracle as O

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo")
    fasta = os.path.join(sys.argv[1], "tests", "golden", "salmonella_enterica_k31_ust.fa.gz")
    path = os.path.join(sys.argv[2], "se.sshash")
    if rank == 0:
        sshash_amd.Dictionary.build(fasta, k=31, m=13, num_threads=2).save(path)
    dist.barrier()
    ora = O.OracleIndex(path)
    lines = gzip.open(os.path.join(sys.argv[1], "tests", "golden", "SRR5833294.10K.fastq.gz"), "rb").read().split(b"\\n")
    reads = lines[1::4][:3001]                                  # ONE set of reads ...
    lo, hi = bench.split_batch(len(reads), world, rank)        # ... read-sharded: contiguous shares, as bench.py --streaming
    names = ("num_kmers", "num_positive_kmers", "num_negative_kmers", "num_invalid_kmers", "num_searches", "num_extensions")
    mine = ora.streaming_query(reads[lo:hi])
    total = torch.tensor([int(mine[f]) for f in names], dtype=torch.int64)
    dist.all_reduce(total, op=dist.ReduceOp.SUM)               # the one reduction: six counters
    if rank == 0:
        whole = ora.streaming_query(reads)
        assert [int(whole[f]) for f in names] == total.tolist(), (whole, total.tolist())
        assert int(total[0]) == sum(max(0, len(r) - 30) for r in reads)
        assert int(total[0]) == int(total[1] + total[2] + total[3]) and int(total[1]) == int(total[4] + total[5])
        print("OK", total.tolist(), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    """
)


def test_read_sharded_streaming_counters_sum_over_two_ranks(tmp_path):
    """bench.py --streaming at N > 1 (BASELINE.json configs[3]: reads sharded, index replicated): every rank queries its contiguous share
    of ONE read set and the six counters are summed with one all_reduce -- with the CPU oracle standing in for the device, the sum over
    two ranks is the report of the whole set (a read is a unit: the streaming state is reset between reads,
    include/streaming_query.hpp:48)."""
    script = tmp_path / "worker.py"
    script.write_text(STREAMING_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2")
    procs = []
    for rank in range(2):
        e = dict(env, RANK=str(rank), LOCAL_RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(tmp_path)], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    assert "OK [" in outs[0]
