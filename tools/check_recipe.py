#!/usr/bin/env python
"""Build the stand-in of a named collection at a fraction of its size and print achieved vs published statistics.
    python tools/check_recipe.py human_k31 0.1 [out.json]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sshash_amd  # noqa: E402
from sshash_amd.repeats import load_recipe, make_recipe_spss, statistics_vs_target  # noqa: E402

name, scale = sys.argv[1], float(sys.argv[2])
r = load_recipe(name)
t0 = time.time()
w, e = make_recipe_spss(name, int(r["reference_bases"] * scale))
t1 = time.time()
d = sshash_amd.Dictionary.build_from_packed(w, e, k=r["k"], m=r["m"], num_threads=0)
t2 = time.time()
cmp = statistics_vs_target(d.bucket_stats(), name)
cmp["seconds"] = {"generate": round(t1 - t0, 1), "build": round(t2 - t1, 1)}
for k, v in cmp.items():
    print(k, v)
if len(sys.argv) > 3:
    json.dump(cmp, open(sys.argv[3], "w"), indent=1)
