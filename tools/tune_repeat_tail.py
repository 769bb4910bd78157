#!/usr/bin/env python
"""Which realisation of the few largest families gives the published tail? (see TAIL_TUNING in calibrate_repeats.py)

Builds every large core class of a recipe ALONE under a dozen class seeds and prints (largest bucket, k-mers in the skew
partitions 5, 6, 7): the realisation of a class depends only on (seed, class seed), so what is printed here is what the
class contributes inside the full stand-in.
    python tools/tune_repeat_tail.py human_k31 [copies core first_seed count]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sshash_amd  # noqa: E402
from sshash_amd.repeats import load_recipe, make_repeat_spss  # noqa: E402

r = load_recipe(sys.argv[1])
seed = 0x5555AAAA
if len(sys.argv) > 2:
    n, core, first, count = (int(v) for v in sys.argv[2:6])
    todo = [({"copies": n, "length": 120 if r["k"] <= 31 else 250, "core": core, "families": 1.0}, range(first, first + count))]
else:
    todo = [(c, range(100, 112)) for c in r["classes"] if "core" in c and c["copies"] >= 11000]
for c, seeds in todo:
    for sd in seeds:
        w, e = make_repeat_spss(int(r["reference_bases"]), k=r["k"], classes=[dict(c, seed=sd)], seed=seed,
                                reference_bases=r["reference_bases"], background=[])
        d = sshash_amd.Dictionary.build_from_packed(w, e, k=r["k"], m=r["m"], num_threads=0)
        s = d.bucket_stats()
        d.close()
        p = list(s["num_kmers_in_skew_partition"]) + [0] * 8
        print(c["copies"], c.get("core"), "class seed", sd, "max bucket", s["max_bucket_size"], "k-mers in partitions 5..7", p[5:8], flush=True)
