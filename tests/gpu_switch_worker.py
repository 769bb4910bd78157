#!/usr/bin/env python
"""Worker of tests/test_gpu_switches.py: one process = one setting of the environment switches that select kernels (they are read
once per process). Every k-mer of a stand-in goes through the id-returning and the is_member instances of the lookup, `launches`
times, in file order and shuffled, forward and reverse-complemented -- a wrong answer that varies from launch to launch (the
gfx950 hazard of HISTORY.md showed up as exactly that) cannot hide behind one lucky launch --, then ASCII input, then a
mixed batch against the CPU oracle. Prints one JSON line; any mismatch is an assertion error.

    python tests/gpu_switch_worker.py <recipe> <bases> <k> <m> <canonical 0|1> <launches>"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import sshash_amd
from oracle import oracle as O
from sshash_amd.repeats import make_recipe_spss
from sshash_amd.synthetic import draw_queries_device, revcomp_device

recipe, bases, k, m, canonical, launches = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), bool(int(sys.argv[5])), int(sys.argv[6])
words, endpoints = make_recipe_spss(recipe, bases, seed=4242)
d = sshash_amd.Dictionary.build_from_packed(words, endpoints, k=k, m=m, canonical=canonical, num_threads=0).to_device(0)
n, W = d.num_kmers(), d.words_per_kmer()
dev = torch.device("cuda", 0)
ids = torch.arange(n, dtype=torch.int64, device=dev)
q = torch.empty((n, W), dtype=torch.int64, device=dev)
d.access_packed_device(0, ids.data_ptr(), n, q.data_ptr())
g = torch.Generator(device=dev)
g.manual_seed(11)
perm = torch.randperm(n, generator=g, device=dev)
cases = [("in order", q, ids), ("reverse complement", revcomp_device(q, k).contiguous(), ids), ("shuffled", q[perm].contiguous(), ids[perm].contiguous())]
out = torch.empty(n, dtype=torch.int64, device=dev)
member = torch.empty(n, dtype=torch.uint8, device=dev)
for rep in range(launches):
    name, qq, want = cases[rep % len(cases)]
    out.fill_(-7)
    member.fill_(7)
    d.lookup_device(0, qq.data_ptr(), n, out.data_ptr())
    d.is_member_device(0, qq.data_ptr(), n, member.data_ptr())
    torch.cuda.synchronize()
    wrong_ids, wrong_member = int((out != want).sum().item()), int((member != 1).sum().item())
    assert wrong_ids == 0 and wrong_member == 0, f"launch {rep} ({name}): {wrong_ids} wrong ids, {wrong_member} wrong is_member answers of {n}"
# ASCII input through the host entry points (their own kernel instances), is_member and ids
if k <= 31:
    sample = q[perm[:2_000_000], 0].cpu().numpy().view(np.uint64)
    text = np.frombuffer(b"ACTG", dtype=np.uint8)[((sample[:, None] >> (2 * np.arange(k, dtype=np.uint64))[None, :]) & np.uint64(3)).astype(np.int64)]
    kmers = [bytes(r) for r in text[:200_000]]
    assert d.is_member(kmers).all()
    assert (d.lookup(kmers).kmer_id == ids[perm[:200_000]].cpu().numpy().view(np.uint64)).all()
# the bench's mix against the oracle
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "w.sshash")
    d.save(path)
    ora = O.OracleIndex(path)
    for negatives in ("random", "mutated"):
        dq = draw_queries_device(d, 0, 1_000_000, 0.5, seed=77, negatives=negatives)
        o2 = torch.empty(1_000_000, dtype=torch.int64, device=dev)
        m2 = torch.empty(1_000_000, dtype=torch.uint8, device=dev)
        d.lookup_device(0, dq.data_ptr(), 1_000_000, o2.data_ptr())
        d.is_member_device(0, dq.data_ptr(), 1_000_000, m2.data_ptr())
        torch.cuda.synchronize()
        want = ora.lookup_ids(dq.cpu().numpy().view(np.uint64), num_threads=8)
        got = o2.cpu().numpy().view(np.uint64)
        assert (got == want).all(), negatives
        assert ((want != np.uint64(0xFFFFFFFFFFFFFFFF)) == (m2.cpu().numpy() == 1)).all(), negatives
st = d.device_stats(0)
print(json.dumps({"ok": True, "kmers": n, "launches": launches, "sk_slots": st["sk_slots"], "directory_sectors": st["directory_sectors"],
                  "switches": {k_: v for k_, v in os.environ.items() if k_.startswith("SSHASH_AMD_")}}))
