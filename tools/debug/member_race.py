#!/usr/bin/env python
"""Debug aid (round 4): WHICH lanes of the is_member instance go wrong without the keep-alive, and what did they store?

Run with SSHASH_AMD_LIBRARY=tools/debug/libsshash_amd_<variant>.so (tools/debug/build_variant.sh). The variants store codes:
  1 = first pass: hit        0 = first pass: miss (or a placeholder nobody rewrote)      7 = never written
  0x41 / 0x40 = the deferred pass's hit / miss      0x80 = first pass: deferred (variant *_first only)
Prints, per launch: histogram of the stored values, the wrong lanes' positions in their wave, how many wrong lanes a wave has, how
many of a wrong wave's other lanes were deferred, and how the wrong set moves from launch to launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import sshash_amd
from sshash_amd.repeats import make_recipe_spss
from sshash_amd.synthetic import revcomp_device

name, bases = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
brief = "brief" in sys.argv[4:]   # one short line per launch; and the same k-mers in shuffled order as well
w, e = make_recipe_spss(name, bases, seed=4242)
canonical = "canonical" in sys.argv[4:]
d = sshash_amd.Dictionary.build_from_packed(w, e, k=31, m=21, canonical=canonical, num_threads=0).to_device(0)
n = d.num_kmers()
dev = torch.device("cuda", 0)
ids = torch.arange(n, dtype=torch.int64, device=dev)
q = torch.empty((n, 1), dtype=torch.int64, device=dev)
d.access_packed_device(0, ids.data_ptr(), n, q.data_ptr())
member = torch.empty(n, dtype=torch.uint8, device=dev)
print("library", os.environ.get("SSHASH_AMD_LIBRARY"), "canonical" if canonical else "regular", {k: os.environ.get(k) for k in ("SSHASH_AMD_INWAVE", "SSHASH_AMD_OVERLAP")}, flush=True)
st = d.device_stats(0)
print({k: st[k] for k in ("sk_keys", "sk_deferred_keys", "sk_heavy_keys", "sk_heavy_kmers")}, "kmers", n, flush=True)
prev = None
cases = [("fwd", q), ("rc", revcomp_device(q, 31).contiguous())]
if brief:
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    cases = [("fwd", q), ("fwd_shuffled", q[torch.randperm(n, generator=g, device=dev)].contiguous())]
for strand, qq in cases:
    for rep in range(reps):
        member.fill_(7)
        torch.cuda.synchronize()
        d.is_member_device(0, qq.data_ptr(), n, member.data_ptr())
        torch.cuda.synchronize()
        m = member.cpu().numpy()
        vals, cnts = np.unique(m, return_counts=True)
        ok = (m == 1) | (m == 0x41)
        bad = np.flatnonzero(~ok)
        line = {"strand": strand, "rep": rep, "values": {hex(int(v)): int(c) for v, c in zip(vals, cnts)}, "wrong": int(bad.size)}
        if bad.size and brief:
            line["lanes_hist_by_16"] = np.bincount((bad & 63) >> 4, minlength=4).tolist()
            if prev is not None and strand == "fwd":
                line["also_wrong_last_time"] = int(np.intersect1d(prev, bad).size)
        elif bad.size:
            lane = bad & 63
            wave = bad >> 6
            uw, per = np.unique(wave, return_counts=True)
            line["wrong_values"] = {hex(int(v)): int(c) for v, c in zip(*np.unique(m[bad], return_counts=True))}
            line["lanes_hist_by_16"] = np.bincount(lane >> 4, minlength=4).tolist()
            line["lane_mod4_hist"] = np.bincount(lane & 3, minlength=4).tolist()
            line["wrong_per_wave_hist"] = {int(a): int(b) for a, b in zip(*np.unique(per, return_counts=True))}
            # the other lanes of a wrong wave: how many were deferred, how many right
            pad = (-n) % 64
            mw = np.concatenate([m, np.full(pad, 1, np.uint8)]).reshape(-1, 64)[uw]
            line["in_wrong_waves"] = {"deferred_lanes": int(((mw == 0x41) | (mw == 0x40) | (mw == 0x80)).sum()), "lanes": int(mw.size)}
            line["wave_in_block_hist"] = np.bincount(uw & 3, minlength=4).tolist()
            line["first"] = bad[:12].tolist()
            runs = np.flatnonzero(np.diff(bad) != 1).size + 1
            line["runs_of_consecutive_wrong"] = int(runs)
            if prev is not None:
                line["also_wrong_last_time"] = int(np.intersect1d(prev, bad).size)
        prev = bad
        print(line, flush=True)
