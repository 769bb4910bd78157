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
