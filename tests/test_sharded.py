"""Minimizer-sharded index. CPU part: the shards partition the
minimizer space and their union answers like the whole dictionary (checked with the oracle). GPU part:
two ranks (gloo group, both on cuda:0) route, exchange with all_to_all, look up and combine."""
from __future__ import annotations

import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

import sshash_amd
from conftest import ROOT, SE_FASTA


@pytest.mark.parametrize("case_name,S", [("case_skew_regular", 2), ("case_skew_canonical", 3)])
def test_shards_partition_the_dictionary(case_name, S, request, tmp_path):
    from oracle import oracle as O

    case = request.getfixturevalue(case_name)
    shards = []
    for r in range(S):
        d = sshash_amd.Dictionary.build(case.fasta, k=case.k, m=case.m, canonical=case.canonical, num_threads=2,
                                        num_shards=S, shard_id=r)
        assert (d.num_shards(), d.shard_id()) == (S, r)
        assert d.num_kmers() == case.gt.num_kmers  # strings stay complete
        p = str(tmp_path / f"shard{r}.sshash")
        d.save(p)
        d2 = sshash_amd.Dictionary.load(p)
        assert (d2.num_shards(), d2.shard_id(), d2.num_minimizers()) == (S, r, d.num_minimizers())
        shards.append((d, O.OracleIndex(p)))
    assert sum(d.num_minimizers() for d, _ in shards) == case.dict.num_minimizers()
    assert all(d.num_minimizers() > 0 for d, _ in shards)
    q = case.queries(3000, 3000, seed=31)
    full = case.oracle.lookup_ids(q)
    per_shard = np.stack([o.lookup_ids(q) for _, o in shards])
    found = per_shard != np.uint64(0xFFFFFFFFFFFFFFFF)
    # a k-mer is answered by its owner; another shard may answer too (its MPHF sends the foreign minimizer to
    # an arbitrary bucket, which in a canonical dictionary can be the reverse-complement minimizer's bucket
    # over the same position) -- never with a different id
    assert (per_shard[found] == np.broadcast_to(full, per_shard.shape)[found]).all()
    assert (found.sum(axis=0) >= (full != np.uint64(0xFFFFFFFFFFFFFFFF))).all()
    combined = np.where(found.any(axis=0), per_shard.min(axis=0), np.uint64(0xFFFFFFFFFFFFFFFF))
    assert (combined == full).all()


def test_bad_shard_arguments(case_skew_regular):
    with pytest.raises(sshash_amd.SSHashError) as e:
        sshash_amd.Dictionary.build(case_skew_regular.fasta, k=31, m=11, num_shards=2, shard_id=2)
    assert e.value.status == 7


WORKER = textwrap.dedent(
    """
    import os, sys
    import numpy as np
    sys.path.insert(0, sys.argv[1])
    import torch, torch.distributed as dist
    import sshash_amd
    from sshash_amd.sharded import ShardedDictionary
    from oracle import oracle as O
    from oracle.ground_truth import GroundTruth, read_fasta_sequences, _revcomp_u64

    fasta, canonical, by = sys.argv[2], sys.argv[3] == "1", sys.argv[4]
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    sd = ShardedDictionary.build(fasta, device=0, by=by, k=31, m=13, canonical=canonical, num_threads=4)
    if by == "table":  # each rank holds its share of the table only, everything else in full
        stats = sd.shard.device_stats(0)
        assert 0 < stats["sk_keys"] < 0.6 * sshash_amd.Dictionary.build(fasta, k=31, m=13, canonical=canonical, num_threads=4).to_device(0).device_stats(0)["sk_keys"]
    # expected ids: the whole dictionary through the oracle
    whole = sshash_amd.Dictionary.build(fasta, k=31, m=13, canonical=canonical, num_threads=4)
    path = f"/tmp/sshash_sharded_test_{os.getpid()}.sshash"
    whole.save(path)
    ora = O.OracleIndex(path)
    os.unlink(path)
    rng = np.random.default_rng(100 + rank)  # every rank looks up its OWN batch
    n = whole.num_kmers()
    ids = rng.integers(0, n, 30000, dtype=np.uint64)
    pos = whole.access_packed(ids)
    pos[::2] = _revcomp_u64(pos[::2], 31)
    neg = rng.integers(0, 1 << 62, 30000, dtype=np.uint64)
    q = np.concatenate([pos, neg])
    rng.shuffle(q)
    got = sd.lookup(q)
    want = ora.lookup_ids(q)
    assert (got == want).all(), f"rank {rank}: sharded ids differ"
    got2 = sd.lookup(q, check_reverse_complement=False)
    assert (got2 == ora.lookup_ids(q, check_rc=False)).all()
    assert sd.lookup(np.zeros(0, dtype=np.uint64)).size == 0  # an empty local batch still takes part in the exchange
    found = int((got != np.uint64(0xFFFFFFFFFFFFFFFF)).sum())
    t = torch.tensor([found]); dist.all_reduce(t)
    if rank == 0:
        print("SHARDED OK", int(t), sd.shard.num_minimizers(), whole.num_minimizers(), flush=True)
    dist.barrier(); dist.destroy_process_group()
    """
)


@pytest.mark.gpu
@pytest.mark.parametrize("canonical,by", [(False, "minimizer"), (True, "minimizer"), (False, "table"), (True, "table")])
def test_two_rank_sharded_lookup_on_gpu(canonical, by, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = 29540 + int(canonical) + 2 * int(by == "table")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, SE_FASTA, "1" if canonical else "0", by],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    assert "SHARDED OK 60000" in outs[0]


MISMATCH_WORKER = textwrap.dedent(
    """
    import os, sys
    import numpy as np
    sys.path.insert(0, sys.argv[1])
    import torch, torch.distributed as dist
    import sshash_amd
    from sshash_amd.sharded import ShardedDictionary

    dist.init_process_group(backend="gloo")
    rank = dist.get_rank()
    torch.cuda.set_device(0)
    sd = ShardedDictionary.build(sys.argv[2], device=0, by="table", k=31, m=13, canonical=False, num_threads=4)
    print("KEY LENGTH", sd.shard.device_stats(0)["sk_key_length"], flush=True)
    try:
        sd.lookup(np.arange(1000, dtype=np.uint64))
        print("NO ERROR", flush=True)
    except sshash_amd.SSHashError as e:
        print("REFUSED:", e, flush=True)
    dist.barrier(); dist.destroy_process_group()
    """
)


@pytest.mark.gpu
def test_ranks_that_elect_table_keys_of_different_lengths_refuse_to_route(tmp_path):
    """SSHASH_AMD_SK_M is read when a replica is built and steers which rank owns a key: ranks started under different environments would
    route keys to ranks that do not hold them, silently. The ranks compare the length at their first collective call."""
    script = tmp_path / "worker.py"
    script.write_text(MISMATCH_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29561", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, SE_FASTA], env=dict(env, RANK=str(r), LOCAL_RANK=str(r), SSHASH_AMD_SK_M=("15", "17")[r]),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o[-3000:]
        assert f"KEY LENGTH {('15', '17')[r]}" in o and "REFUSED:" in o and "SSHASH_AMD_SK_M must be the same on every rank" in o and "NO ERROR" not in o, o[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("case_name", ["case_skew_regular", "case_k63_canonical"])
def test_a_table_shard_answers_every_query_on_its_own(case_name, request):
    """A replica whose super-k-mer table holds one third of the keys must still answer ANY query correctly: a key
    of another shard takes the complete path (slower, never wrong)."""
    case = request.getfixturevalue(case_name)
    d = sshash_amd.Dictionary.load(case.index_path).to_device(0, table_shards=3, table_shard_id=1)
    full = case.dict.to_device(0).device_stats()["sk_keys"]
    mine = d.device_stats()["sk_keys"]
    assert 0 < mine < full
    n = case.gt.num_kmers
    allq = case.gt.kmers(np.arange(n))
    assert (d.lookup(allq).kmer_id == np.arange(n, dtype=np.uint64)).all()
    q = case.queries(3000, 3000, seed=9)
    want = case.oracle.lookup_ids(q)
    assert (d.lookup(q).kmer_id == want).all()
    assert (d.is_member(q) == (want != np.uint64(0xFFFFFFFFFFFFFFFF))).all()
    reads = ["".join(case.sequences[i]) for i in range(0, len(case.sequences), 5)]
    got, ref = d.streaming_query(reads), case.oracle.streaming_query(reads)
    assert [getattr(got, f) for f in ref] == list(ref.values())


@pytest.mark.gpu
def test_sharded_lookup_from_cpp_with_callbacks_and_over_rccl():
    """tests/cpp/check_sharded.cpp: two ranks as host threads over sshash_sharded_lookup_device with their own exchange
    callbacks (minimizer shards, then table shards), and sshash_sharded_lookup_rccl over a real RCCL communicator."""
    binary = os.path.join(ROOT, "tests", "cpp", "check_sharded")
    if not os.path.exists(binary):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "sshash_amd", "csrc"), "tools"])
    p = subprocess.run([binary, SE_FASTA, "31", "13"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "EVERYTHING OK!" in p.stdout


NCCL_WORKER = textwrap.dedent(
    """
    import os, sys
    import numpy as np
    sys.path.insert(0, sys.argv[1])
    import torch, torch.distributed as dist
    import sshash_amd
    from sshash_amd.sharded import ShardedDictionary
    from oracle.ground_truth import _revcomp_u64

    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))  # RCCL
    whole = sshash_amd.Dictionary.build(sys.argv[2], k=31, m=13, num_threads=4).to_device(local)
    for by in ("minimizer", "table"):
        sd = ShardedDictionary.build(sys.argv[2], device=local, by=by, k=31, m=13, num_threads=4)
        rng = np.random.default_rng(5 + rank)
        ids = rng.integers(0, whole.num_kmers(), 20000, dtype=np.uint64)
        pos = whole.access_packed(ids)
        pos[::2] = _revcomp_u64(pos[::2], 31)
        q = np.concatenate([pos, rng.integers(0, 1 << 62, 20000, dtype=np.uint64)])
        got = sd.lookup(q)
        assert (got == whole.lookup(q).kmer_id).all(), by
        assert sd.lookup(np.zeros(0, dtype=np.uint64)).size == 0
    if rank == 0:
        print("NCCL SHARDED OK", world, flush=True)
    dist.barrier(); dist.destroy_process_group()
    """
)


@pytest.mark.gpu
def test_sharded_lookup_over_the_nccl_backend(tmp_path):
    """The RCCL branch of sshash_amd/sharded.py (all_to_all_single on device tensors) under torch.distributed.run: as many
    ranks as there are GPUs (one on the single-GPU test box -- the exchange with oneself still goes through RCCL)."""
    import torch

    n = max(1, min(torch.cuda.device_count(), 8))
    script = tmp_path / "nccl_worker.py"
    script.write_text(NCCL_WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
                        "--master-port", "29577", str(script), ROOT, SE_FASTA], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    assert f"NCCL SHARDED OK {n}" in p.stdout


BUDGET_WORKER = textwrap.dedent(
    """
    import os, sys, time
    import numpy as np
    sys.path.insert(0, sys.argv[1])
    import torch, torch.distributed as dist
    import sshash_amd
    from sshash_amd.sharded import ShardedDictionary
    from sshash_amd.synthetic import draw_queries_device
    from oracle import oracle as O

    index_path, by, budget, n_queries = sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    whole = sshash_amd.Dictionary.load(index_path)
    # the queries are drawn from an unrestricted replica of the whole dictionary, one rank at a time (they time-share ONE GPU)
    for turn in range(world):
        if turn == rank:
            whole.to_device(0)
            dq = draw_queries_device(whole, 0, n_queries, 0.5, seed=1000 + rank).clone()
            whole.close()
            torch.cuda.synchronize(); torch.cuda.empty_cache()
        dist.barrier()
    os.environ["SSHASH_AMD_HBM_BUDGET"] = str(budget)
    # 1. a whole replica does not fit the budget ...
    whole = sshash_amd.Dictionary.load(index_path)
    try:
        whole.to_device(0)
        stats = whole.device_stats(0)
        # (a budget between "everything but the table" and "everything": the replica is there, without its table, and says why)
        assert stats["sk_slots"] == 0 and stats["sk_absent_reason"] == "not enough free HBM", stats
        fits_without_table = True
    except sshash_amd.SSHashError as e:
        assert "SSHASH_AMD_HBM_BUDGET" in str(e), str(e)
        fits_without_table = False
    whole.close()
    # 2. ... its partition over the ranks does
    if by == "table":
        sd = ShardedDictionary(sshash_amd.Dictionary.load(index_path), 0, by="table")
        stats = sd.shard.device_stats(0)
        assert stats["sk_slots"] > 0 and stats["sk_absent_reason"] is None and stats["bytes"] <= budget, stats
    else:
        shard = sshash_amd.Dictionary.load(sys.argv[6] % rank)
        sd = ShardedDictionary(shard, 0, by="minimizer")
        stats = shard.device_stats(0)
        assert stats["sk_absent_reason"] == "minimizer shard" and stats["bytes"] <= budget, stats
    t0 = time.time()
    got = sd.lookup_device(dq).cpu().numpy().view(np.uint64)
    dt = time.time() - t0
    want = O.OracleIndex(index_path).lookup_ids(dq.cpu().numpy().view(np.uint64), num_threads=max(1, (os.cpu_count() or 4) // world))
    assert (got == want).all(), f"rank {rank}: {int((got != want).sum())} ids differ from the oracle"
    found = torch.tensor([int((got != np.uint64(0xFFFFFFFFFFFFFFFF)).sum())]); dist.all_reduce(found)
    if rank == 0:
        print("BUDGET OK", by, world, int(found), "fits_without_table" if fits_without_table else "does_not_fit", f"{dt:.2f}s", flush=True)
    dist.barrier(); dist.destroy_process_group()
    """
)


@pytest.mark.gpu
@pytest.mark.parametrize("by", ["table", "minimizer"])
def test_a_dictionary_that_needs_sharding(by, tmp_path):
    """BASELINE.json configs[4] on the one-GPU box: SSHASH_AMD_HBM_BUDGET caps what a replica may hold, so that
    a C2-like dictionary (the S. enterica recipe at 1/12 of its size: 75 M k-mers) CANNOT get a whole replica -- the table is
    refused, or the replica altogether, and device_stats says why -- and must be partitioned: FOUR ranks time-sharing the GPU,
    table shards or minimizer shards, 10^7 queries of the bench mix per rank through the routed lookup, every id against the oracle."""
    from sshash_amd.repeats import make_recipe_spss

    world, n_queries = 4, 10_000_000
    words, ends = make_recipe_spss("se_k31", 115_000_000, seed=3)
    whole = sshash_amd.Dictionary.build_from_packed(words, ends, k=31, m=21, num_threads=0)
    index = str(tmp_path / "whole.sshash")
    whole.save(index)
    shard_pattern = str(tmp_path / "shard%d.sshash")
    if by == "minimizer":
        for r in range(world):
            s = sshash_amd.Dictionary.build_from_packed(words, ends, k=31, m=21, num_threads=0, num_shards=world, shard_id=r)
            s.save(shard_pattern % r)
            s.close()
    # budgets from the layout itself: a whole replica (table included) measured without a limit
    whole.to_device(0)
    st = whole.device_stats(0)
    whole.close()
    full, table = st["bytes"], st["sk_bytes"]
    # table shards: everything but the table in full + a quarter of the table (+ slack) fits, the whole table does not;
    # minimizer shards: not even the table-less replica fits (strings + a quarter of the minimizer structures do)
    if by == "table":
        budget = (full - table) + table // world + table // 8
    else:
        os.environ["SSHASH_AMD_SKTABLE"] = "0"
        try:
            w2 = sshash_amd.Dictionary.load(index).to_device(0)
            without_table = w2.device_bytes(0)
            w2.close()
        finally:
            del os.environ["SSHASH_AMD_SKTABLE"]
        s0 = sshash_amd.Dictionary.load(shard_pattern % 0).to_device(0)
        one_shard = s0.device_bytes(0)
        s0.close()
        assert one_shard * 1.1 < without_table, (one_shard, without_table)
        budget = int((one_shard * 1.1 + without_table) / 2)
    script = tmp_path / "budget_worker.py"
    script.write_text(BUDGET_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29560 + (by == "table")), WORLD_SIZE=str(world))
    env.pop("SSHASH_AMD_HBM_BUDGET", None)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, index, by, str(budget), str(n_queries), shard_pattern],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = [p.communicate(timeout=1500)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    assert f"BUDGET OK {by} {world}" in outs[0]
    assert ("fits_without_table" if by == "table" else "does_not_fit") in outs[0], outs[0][-500:]
