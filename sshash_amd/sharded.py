"""Minimizer-sharded dictionary: one shard of the sparse-and-skew index per GPU, queries routed with
an all-to-all (BASELINE.json config 5: an index larger than one GPU's HBM).

The replicated mode (bench.py) needs no collective. This mode is for dictionaries whose
minimizer-side structures (directory / MPHF / control codewords / bucket lists, ~70 % of the index) do
not fit one HBM: rank r holds only the buckets of the minimizers with ``shard_of_minimizer(mu, R) == r``
(``Dictionary.build(..., num_shards=R, shard_id=r)``), plus the complete strings.

Lookup of a local batch on rank r (all device-side except the split-size exchange):

 1. route      ``sshash_route_bucket_device`` twice: owner of the forward minimizer and of the reverse-complement
               minimizer of every query (equal for canonical dictionaries); count the messages per owner, then
               scatter them into contiguous per-owner regions (LDS histogram + one reservation per workgroup and
               owner) together with the index of the query each one is about;
 2. exchange   one message per (query, distinct owner): the packed k-mer; ``all_to_all_single`` over
               RCCL (xGMI) -- 8*W bytes per message out, 8 bytes back;
 3. lookup     every rank runs the ordinary batched lookup on what it received: a probe whose minimizer
               lives on another shard simply misses (its fingerprint is not there);
 4. return     ids travel back with the inverse all-to-all;
 5. combine    ``sshash_route_combine_device``: a reply that found the k-mer settles its query (a k-mer occurs
               once in the strings, so two owners that both find it return the same id).

Results are identical to the unsharded dictionary: ids derive from string offsets, which are global.

A second partitioning, by = "table": every rank keeps the complete reference structures and strings but only its
share of the super-k-mer table (about three quarters of a replica's HBM); queries are routed by their table key,
which is strand-symmetric, so a query has ONE owner and the exchange carries one message per query.
The five steps are ONE call of the C ABI, ``sshash_sharded_lookup_device`` (sshash_amd/csrc/sharded.cpp), which takes
the exchange as two callbacks; this module supplies them from ``torch.distributed`` (``all_to_all_single`` over
RCCL; a ``gloo`` group -- the tests -- stages the payloads through host memory). A C++ host hands its own
exchange, or an ``ncclComm_t`` to ``sshash_sharded_lookup_rccl``, and needs nothing of this file.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from ._binding import INVALID_U64, Dictionary


class ShardedDictionary:
    """by = "minimizer": `shard` is rank r's minimizer shard of the index (built with num_shards/shard_id).
    by = "table": `shard` is the COMPLETE dictionary; only the device's super-k-mer table -- the largest structure
    in HBM -- is partitioned: rank r builds the slots of the keys it owns, and a query goes to the owner of its
    (strand-symmetric) table key: one message per query instead of up to two."""

    def __init__(self, shard: Dictionary, device: int, group=None, by: str = "minimizer"):
        import torch
        import torch.distributed as dist

        if by not in ("minimizer", "table"):
            raise ValueError("by must be 'minimizer' or 'table'")
        self.shard = shard
        self.by = by
        self.device = int(device)
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self._on_host = dist.get_backend(group) == "gloo"
        self._dev = torch.device("cuda", self.device)
        if by == "table":
            if shard.num_shards() != 1:
                raise ValueError("table sharding partitions the device table of a complete dictionary")
            shard.to_device(self.device, table_shards=self.world, table_shard_id=self.rank)
        else:
            if shard.num_shards() != self.world or shard.shard_id() != self.rank:
                raise ValueError(f"rank {self.rank}/{self.world} was given shard {shard.shard_id()}/{shard.num_shards()}")
            shard.to_device(self.device)

    @classmethod
    def build(cls, input_filename: str, device: int, group=None, by: str = "minimizer", **build_kwargs) -> "ShardedDictionary":
        import torch.distributed as dist

        if by == "table":
            return cls(Dictionary.build(input_filename, **build_kwargs), device, group, by)
        shard = Dictionary.build(input_filename, num_shards=dist.get_world_size(group), shard_id=dist.get_rank(group),
                                 **build_kwargs)
        return cls(shard, device, group)

    # -- the exchange: torch.distributed behind the two callbacks of sshash_sharded_lookup_device ------------------
    class _DevicePointer:
        """A raw device pointer as something torch.as_tensor accepts (the CUDA array interface)."""

        def __init__(self, ptr: int, nbytes: int):
            self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}

    def _tensor(self, ptr: int, nbytes: int):
        import torch

        if nbytes == 0:
            return torch.empty(0, dtype=torch.uint8, device=self._dev)
        return torch.as_tensor(self._DevicePointer(ptr, nbytes), device=self._dev)

    def _exchange_counts(self, send_counts):
        import torch
        import torch.distributed as dist

        if self.world == 1:
            return [int(x) for x in send_counts]
        s = torch.tensor(send_counts, dtype=torch.int64, device="cpu" if self._on_host else self._dev)
        r = torch.empty_like(s)
        dist.all_to_all_single(r, s, group=self.group)
        return [int(x) for x in r.tolist()]

    def _exchange_data(self, send_ptr, send_counts, recv_ptr, recv_counts, elem_bytes, stream):
        """all-to-all-v of device buffers: all_to_all_single over RCCL (xGMI); a gloo group (tests) stages through host
        memory. The library's launches are on `stream`, which is torch's current stream (lookup_device passes it)."""
        import torch
        import torch.distributed as dist

        send = self._tensor(send_ptr, sum(send_counts) * elem_bytes)
        recv = self._tensor(recv_ptr, sum(recv_counts) * elem_bytes)
        s_split = [c * elem_bytes for c in send_counts]
        r_split = [c * elem_bytes for c in recv_counts]
        if self.world == 1:  # a group of one: what is sent is what arrives
            recv.copy_(send)
        elif self._on_host:
            r = torch.empty(recv.numel(), dtype=torch.uint8)
            dist.all_to_all_single(r, send.cpu(), r_split, s_split, group=self.group)
            recv.copy_(r)
        else:
            dist.all_to_all_single(recv, send, r_split, s_split, group=self.group)

    # -- lookup ----------------------------------------------------------------------------------------
    def lookup_device(self, d_kmers, check_reverse_complement: bool = True):
        """d_kmers: int64 CUDA tensor of n*W packed words on this rank's device -> int64 CUDA tensor of n ids
        (bit pattern of the uint64 ids; INVALID_U64 == -1). Collective: every rank of the group calls it (an empty
        local batch still takes part in the exchange). Route, lookup, return and combine run inside
        sshash_sharded_lookup_device (sshash_amd/csrc/sharded.cpp); only the exchange comes from here."""
        import torch

        W = self.shard.words_per_kmer()
        n = d_kmers.numel() // W
        stream = torch.cuda.current_stream(self._dev).cuda_stream
        out = torch.empty(max(n, 1), dtype=torch.int64, device=self._dev)
        d_kmers = d_kmers.contiguous()
        self.shard.sharded_lookup_device(self.device, self.world, self.by == "table", d_kmers.data_ptr() if n else 0, n,
                                         out.data_ptr(), self._exchange_counts, self._exchange_data,
                                         check_reverse_complement=check_reverse_complement, stream=stream)
        return out[:n]

    def lookup(self, kmers: np.ndarray, check_reverse_complement: bool = True) -> np.ndarray:
        """Host convenience wrapper: packed uint64 words in, uint64 ids out."""
        import torch

        a = np.ascontiguousarray(kmers, dtype=np.uint64)
        d = torch.from_numpy(a.view(np.int64)).to(self._dev)
        return self.lookup_device(d, check_reverse_complement).cpu().numpy().view(np.uint64)
