#!/usr/bin/env python
"""Debug aid: is_member / lookup over EVERY k-mer of a stand-in, several launches, under the current environment switches.

Written in round 3 for the wrong is_member answers that round 4 traced to a hardware hazard (a 64-bit shift by the
last VGPR of the wave's allocation; tools/isa_guard.py keeps it out of the library). To see it again, build the library WITHOUT the guard
(tools/debug/build_variant.sh unguarded: a plain hipcc -c of engine.hip) and run
    SSHASH_AMD_LIBRARY=$PWD/tools/debug/libsshash_amd_unguarded.so python tools/debug/member_mismatch.py se_k31 20000000
(tools/debug/member_race.py prints more: which lanes, what they stored, how the wrong set moves from launch to launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sshash_amd
from sshash_amd.repeats import make_recipe_spss
from sshash_amd.synthetic import revcomp_device

name, bases = sys.argv[1], int(sys.argv[2])
w, e = make_recipe_spss(name, bases, seed=4242)
d = sshash_amd.Dictionary.build_from_packed(w, e, k=31, m=21, num_threads=0).to_device(0)
n = d.num_kmers()
dev = torch.device("cuda", 0)
ids = torch.arange(n, dtype=torch.int64, device=dev)
q = torch.empty((n, 1), dtype=torch.int64, device=dev)
d.access_packed_device(0, ids.data_ptr(), n, q.data_ptr())
out = torch.empty(n, dtype=torch.int64, device=dev)
member = torch.empty(n, dtype=torch.uint8, device=dev)
print({k: os.environ.get(k) for k in ("SSHASH_AMD_INWAVE", "SSHASH_AMD_SKTABLE")}, d.device_stats(0))
for rep in range(4):
    for qq in (q, revcomp_device(q, 31).contiguous()):
        member.fill_(7)
        d.lookup_device(0, qq.data_ptr(), n, out.data_ptr())
        d.is_member_device(0, qq.data_ptr(), n, member.data_ptr())
        torch.cuda.synchronize()
        bad_ids = int((out != ids).sum().item())
        bad = (member != 1).nonzero().flatten()
        vals = member[bad][:8].tolist()
        print(rep, "ids wrong", bad_ids, "member wrong", int(bad.numel()), "first", bad[:6].tolist(), "values", vals, flush=True)
