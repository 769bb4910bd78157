#!/usr/bin/env python
"""Fit the family-class amounts of sshash_amd/repeats.py to the bucket statistics the reference printed for a real
collection (benchmarks/results-10-11-25/k31/regular-build.log), and write sshash_amd/recipes/<name>.json.

Method: every candidate class (copies, length, divergence) is generated ALONE, indexed with this repo's builder, and its
statistics per family are read off sshash_bucket_stats (classes share no sequence, so a mixture's statistics are the sum
of its classes'). Two background classes (short and long random strings) carry the singleton buckets, the string count and
the base count. The amounts x >= 0 minimise the relative error over the published numbers (scipy NNLS):
buckets of exactly 2..16 positions, buckets / positions of 17..64, skew-index buckets / positions / k-mers per partition,
minimizers, positions, strings, k-mers. The recipe is then built once at `--check-scale` of the full size and the
achieved statistics are stored next to the targets.

    python tools/calibrate_repeats.py human_k31 [--check-scale 0.1]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

# ---- published statistics (reference benchmarks/results-10-11-25/k31/regular-build.log) -----------------------------
TARGETS = {
    "human_k31": {  # block from line 336: human.k31.eulertigs.fa.gz, k=31 m=21 regular
        "source": "benchmarks/results-10-11-25/k31/regular-build.log:336-510", "k": 31, "m": 21,
        "num_kmers": 2505678680, "num_strings": 10250465, "num_bases": 2813192630,
        "num_minimizers": 386687326, "num_minimizer_positions": 423023926,
        "num_buckets_larger_than_1_not_in_skew_index": 10816752, "num_minimizer_positions_of_buckets_larger_than_1": 40422973,
        "num_buckets_in_skew_index": 42372, "num_minimizer_positions_of_buckets_in_skew_index": 6772751,
        "num_kmers_in_skew_index": 32063746, "max_bucket_size": 22972,
        "num_kmers_in_skew_partition": [11807213, 8389556, 5343660, 3076413, 1855446, 1008178, 375770, 207510],
        "bucket_percent": [97.1918, 1.69205, 0.442612, 0.201662, 0.114623, 0.0734475, 0.0506719, 0.0368406, 0.0280203, 0.0218197,
                           0.0175108, 0.0142045, 0.0116337, 0.00980818, 0.00832637, 0.00717712],
    },
    "human_k63": {  # benchmarks/results-10-11-25/k63/regular-build.log, block from line 323: human.k63.eulertigs.fa.gz, k=63 m=25 regular
        "source": "benchmarks/results-10-11-25/k63/regular-build.log:323-496", "k": 63, "m": 25,
        "num_kmers": 2771316093, "num_strings": 2642917, "num_bases": 2935176947,
        "num_minimizers": 122838669, "num_minimizer_positions": 140756047,
        "num_buckets_larger_than_1_not_in_skew_index": 3097190, "num_minimizer_positions_of_buckets_larger_than_1": 12724460,
        "num_buckets_in_skew_index": 28203, "num_minimizer_positions_of_buckets_in_skew_index": 8318311,
        "num_kmers_in_skew_index": 145458128, "max_bucket_size": 147936,
        "num_kmers_in_skew_partition": [25196923, 21919654, 19634878, 18051454, 17018125, 14085569, 9296403, 20255122],
        "bucket_percent": [97.4557, 1.46862, 0.405135, 0.185188, 0.10523, 0.0680771, 0.0480598, 0.0352926, 0.0271502, 0.0217871,
                           0.0176866, 0.014607, 0.0121411, 0.0105985, 0.00909811, 0.00794457],
    },
    "se_k31": {  # block from line 1834: se.k31.eulertigs.fa.gz (S. enterica pangenome), k=31 m=21 regular
        "source": "benchmarks/results-10-11-25/k31/regular-build.log:1834-2022", "k": 31, "m": 21,
        "num_kmers": 894310084, "num_strings": 16440873, "num_bases": 1387536274,
        "num_minimizers": 126246665, "num_minimizer_positions": 162006751,
        "num_buckets_larger_than_1_not_in_skew_index": 14059268, "num_minimizer_positions_of_buckets_larger_than_1": 48164669,
        "num_buckets_in_skew_index": 8266, "num_minimizer_positions_of_buckets_in_skew_index": 1662951,
        "num_kmers_in_skew_index": 6466768, "max_bucket_size": 36894,
        "num_kmers_in_skew_partition": [2254325, 1183762, 885561, 591648, 450833, 373731, 338406, 388502],
        "bucket_percent": [88.8571, 6.64024, 1.83049, 0.817717, 0.475524, 0.319677, 0.233881, 0.178737, 0.138965, 0.107245,
                           0.0823246, 0.0632579, 0.0483173, 0.0365673, 0.027916, 0.0218604],
    },
}


# The largest buckets belong to a handful of families, and which of a family's core m-mers become buckets is a matter of
# their hash: the expected amounts above leave the top partitions (> 4096 positions) and the largest bucket to luck. The
# realisation of a class depends only on (seed, class seed) (repeats.py), so the few top classes were built alone under a
# dozen class seeds each (tools/tune_repeat_tail.py) and the combination closest to the published tail was written down here.
TAIL_TUNING = {
    "human_k31": {
        "class_seeds": {(11264, 21): 106, (15930, 21): 105, (22528, 21): 111},
        "extra_classes": [{"copies": 30000, "length": 120, "core": 21, "families": 1.0, "seed": 305,
                           "note": "the largest bucket: one family whose core wins its windows (max bucket 21 087 alone)"}],
    },
    "se_k31": {"class_seeds": {(11264, 26): 104, (22528, 23): 103}},
    "human_k63": {
        "extra_classes": [{"copies": 160000, "length": 250, "core": 45, "families": 1.0, "seed": 100,
                           "note": "the largest bucket: 148 015 alone, 5.0 M k-mers in the last skew partition"}],
        # ... which the fit had given to the 63 719-copy cores (1.9 M k-mers of that partition per family)
        "families_delta": {(63719, 45): -2.5},
    },
}


def target_vector(t):
    """the rows the fit works on, as a dict name -> value"""
    hist = [p / 100.0 * t["num_minimizers"] for p in t["bucket_percent"]]
    rows = {}
    for s in range(2, 17):
        rows[f"buckets_{s}"] = hist[s - 1]
    small_b = sum(hist[1:])
    small_p = sum((s + 1) * hist[s] for s in range(1, 16))
    rows["buckets_17_64"] = t["num_buckets_larger_than_1_not_in_skew_index"] - small_b
    rows["positions_17_64"] = t["num_minimizer_positions_of_buckets_larger_than_1"] - small_p
    rows["skew_buckets"] = t["num_buckets_in_skew_index"]
    rows["skew_positions"] = t["num_minimizer_positions_of_buckets_in_skew_index"]
    for p, v in enumerate(t["num_kmers_in_skew_partition"]):
        rows[f"skew_kmers_{p}"] = v
    rows["minimizers"] = t["num_minimizers"]
    rows["positions"] = t["num_minimizer_positions"]
    rows["strings"] = t["num_strings"]
    rows["kmers"] = t["num_kmers"]
    return rows


def stats_vector(s):
    hist = s["buckets_with_n_positions"]
    rows = {}
    for n in range(2, 17):
        rows[f"buckets_{n}"] = hist[n - 1]
    small_b = sum(hist[1:])
    small_p = sum((n + 1) * hist[n] for n in range(1, 16))
    rows["buckets_17_64"] = s["num_buckets_larger_than_1_not_in_skew_index"] - small_b
    rows["positions_17_64"] = s["num_minimizer_positions_of_buckets_larger_than_1"] - small_p
    rows["skew_buckets"] = s["num_buckets_in_skew_index"]
    rows["skew_positions"] = s["num_minimizer_positions_of_buckets_in_skew_index"]
    part = list(s["num_kmers_in_skew_partition"]) + [0] * 8
    for p in range(8):
        rows[f"skew_kmers_{p}"] = part[p]
    rows["minimizers"] = s["num_minimizers"]
    rows["positions"] = s["num_minimizer_positions"]
    rows["strings"] = s["num_strings"]
    rows["kmers"] = s["num_kmers"]
    return rows


def candidate_classes_k63(max_bucket):
    """the same idea at k = 63, m = 25: longer copies (a string must hold 63-mers), divergences at which 25-mers survive in many
    copies and 63-mers in few, cores of 25..60 bases; the diverged classes stop at 128 k copies (the largest buckets are left
    to the cores: a k = 63 family of a million 600-base copies is a third of the collection)"""
    copies = [2, 3, 4, 6, 8]
    n = 11.0
    while n < 300000:
        copies.append(int(round(n)))
        n *= 2 ** 0.5
    out = []
    for c in copies:
        if c <= 8:
            opts = [(1000, 0.002), (1000, 0.01), (2000, 0.03), (2000, 0.06)]
        elif c <= 256:
            opts = [(400, 0.015), (600, 0.04), (800, 0.08)]
        elif c <= 130000:
            opts = [(300, 0.03), (400, 0.06), (600, 0.10)]
        else:
            opts = []
        for L, dv in opts:
            out.append({"copies": c, "length": L, "divergence": dv})
        for core in (25, 31, 45, 60):
            out.append({"copies": c, "length": 250, "core": core})
            if c <= 16:
                out.append({"copies": c, "length": 3000, "core": core})
    return out


def candidate_classes(max_bucket):
    """(copies, length, divergence) grid: copies in steps of about sqrt(2); per size range three (length, divergence)
    regimes -- near-identical copies (variant bubbles: short private strings around each substitution), diverged copies,
    old families whose copies share m-mers but hardly any k-mer (they stay whole strings)"""
    copies = [2, 3, 4, 6, 8]
    n = 11.0
    while n < 1.5e6:
        copies.append(int(round(n)))
        n *= 2 ** 0.5
    out = []
    for c in copies:
        if c <= 8:
            opts = [(300, 0.005), (300, 0.02), (600, 0.06), (1000, 0.04), (1000, 0.10)]
        elif c <= 256:
            opts = [(200, 0.03), (300, 0.08), (400, 0.15)]
        else:
            opts = [(150, 0.06), (200, 0.12), (300, 0.18)]
            if c >= 90000:
                opts += [(200, 0.09)]
            opts += [(50, 0.09), (80, 0.12)]
        for L, dv in opts:
            out.append({"copies": c, "length": L, "divergence": dv})
        # short exact repeats in unrelated contexts: the few largest buckets of a real collection belong to a handful
        # of m-mers, not to the hundred consensus minimizers of a long diverged family
        for core in (21, 23, 26, 29):
            if c > 40000:
                break
            out.append({"copies": c, "length": 120, "core": core})
            if c <= 16:  # the same inside long strings: repeats that cost no extra string ends
                out.append({"copies": c, "length": 2000, "core": core})
    return out


def measure(cls, k, m, raw_budget=6e6, seed=99):
    import sshash_amd
    from sshash_amd.repeats import make_repeat_spss

    per_family = cls["copies"] * cls["length"]
    if "core" in cls:
        raw_budget *= 5  # which cores become large buckets is a matter of their hash: average over more families
    F = max(1, int(raw_budget // per_family))
    c = dict(cls, families=F)
    w, e = make_repeat_spss(1, k=k, classes=[c], reference_bases=1, seed=seed)
    d = sshash_amd.Dictionary.build_from_packed(w, e, k=k, m=m, num_threads=0)
    s = d.bucket_stats()
    d.close()
    v = stats_vector(s)
    return {kk: vv / F for kk, vv in v.items()}, s["max_bucket_size"], s["num_bases"] / F


def measure_background(mean_len, k, m, bases=20_000_000, seed=7):
    import sshash_amd
    from sshash_amd.repeats import make_repeat_spss

    w, e = make_repeat_spss(bases, k=k, classes=[], mean_len=mean_len, seed=seed)
    d = sshash_amd.Dictionary.build_from_packed(w, e, k=k, m=m, num_threads=0)
    s = d.bucket_stats()
    d.close()
    v = stats_vector(s)
    return {kk: vv / s["num_bases"] for kk, vv in v.items()}  # per base


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name", choices=sorted(TARGETS))
    ap.add_argument("--check-scale", type=float, default=0.1)
    ap.add_argument("--feedback", default=None,
                    help="a bench.py JSON line of the CURRENT recipe at full size: every target row is multiplied by target / achieved "
                         "before the fit (one step of a fixed-point iteration: the per-class vectors are measured on one or a few "
                         "families each, the stand-in holds other realisations of them)")
    ap.add_argument("--freeze-cores", action="store_true",
                    help="with --feedback: the core classes keep the amounts (and class seeds) of the current recipe -- the tail they "
                         "realise was chosen seed by seed and must not move -- and only the diverged families and the background are refitted")
    ap.add_argument("--strings-weight", type=float, default=None,
                    help="weight of the number of strings in the fit (default 4 like the other totals at k <= 31; 1 at k = 63, where "
                         "the published collection's strings are long unitigs holding many repeated m-mers each and the families' "
                         "are one copy each: the bucket classes matter to the lookup, the string count hardly does)")
    ap.add_argument("--cache", default=os.path.join(ROOT, "tools", "calibrate_repeats_cache.json"))
    args = ap.parse_args()
    from scipy.optimize import nnls

    t = TARGETS[args.name]
    k, m = t["k"], t["m"]
    tv = target_vector(t)
    names = list(tv)
    aim = dict(tv)  # what the fit aims at: the published numbers, corrected by what the last stand-in achieved
    if args.feedback:
        line = json.loads(open(args.feedback).read().strip().splitlines()[-1])
        st = line["config"]["index_statistics"]
        ach = {"num_kmers": st["num_kmers"]["achieved"], "num_strings": st["num_strings"]["achieved"],
               "num_minimizers": st["num_minimizers"]["achieved"], "num_minimizer_positions": st["num_minimizer_positions"]["achieved"],
               "num_buckets_larger_than_1_not_in_skew_index": st["num_buckets_larger_than_1_not_in_skew_index"]["achieved"],
               "num_minimizer_positions_of_buckets_larger_than_1": st["num_minimizer_positions_of_buckets_larger_than_1"]["achieved"],
               "num_buckets_in_skew_index": st["num_buckets_in_skew_index"]["achieved"],
               "num_minimizer_positions_of_buckets_in_skew_index": st["num_minimizer_positions_of_buckets_in_skew_index"]["achieved"],
               "num_kmers_in_skew_partition": [st[f"num_kmers_in_skew_partition_{p}"]["achieved"] for p in range(8)],
               "buckets_with_n_positions": [st[f"buckets_with_{n}_positions"]["achieved"] for n in range(1, 17)]}
        av = stats_vector(ach)
        scale = st["scale"]
        for n in names:
            if av[n] > 0:
                ratio = tv[n] * scale / av[n]
                aim[n] = tv[n] * min(1.5, max(0.67, ratio))
        # the recipe being corrected is the starting point: its own aim is carried along (corrections accumulate)
        prev = os.path.join(ROOT, "sshash_amd", "recipes", args.name + ".json")
        if os.path.exists(prev):
            old = json.load(open(prev)).get("aim")
            if old:
                for n in names:
                    aim[n] = old[n] * (aim[n] / tv[n])
    cache = {}
    if os.path.exists(args.cache):
        cache = json.load(open(args.cache))
    classes = candidate_classes_k63(t["max_bucket_size"]) if k > 31 else candidate_classes(t["max_bucket_size"])
    cols, meta = [], []
    t0 = time.time()
    for c in classes:
        key = f"{k}-{m}-{c['copies']}-{c['length']}-" + (f"core{c['core']}" if "core" in c else f"{c['divergence']}")
        if key not in cache:
            v, mx, bases = measure(c, k, m)
            cache[key] = {"v": v, "max": mx, "bases": bases}
            json.dump(cache, open(args.cache, "w"))
            print(f"[{time.time() - t0:6.1f}s] {key}: max bucket {mx}, bases/family {bases:.0f}", file=sys.stderr)
        if cache[key]["max"] > 1.15 * t["max_bucket_size"]:
            continue  # this class alone would exceed the largest published bucket
        cols.append([cache[key]["v"][n] for n in names])
        meta.append(dict(c, max_bucket=cache[key]["max"], bases_per_family=cache[key]["bases"]))
    backgrounds = [80.0, 400.0, 4000.0] if k <= 31 else [150.0, 1000.0, 20000.0]
    for ml in backgrounds:
        key = f"{k}-{m}-bg-{ml}"
        if key not in cache:
            cache[key] = {"v": measure_background(ml, k, m)}
            json.dump(cache, open(args.cache, "w"))
        cols.append([cache[key]["v"][n] * 1e6 for n in names])  # unit: 10^6 bases
        meta.append({"background_mean_len": ml})
    A = np.array(cols, dtype=np.float64).T  # rows x classes
    b = np.array([aim[n] for n in names], dtype=np.float64)
    # relative errors; the totals weigh more (they are what "the same size" means), the thin tail rows less
    wgt = np.ones(len(names))
    for i, n in enumerate(names):
        if n in ("minimizers", "positions", "strings", "kmers"):
            wgt[i] = 4.0
        if n == "strings":
            wgt[i] = args.strings_weight if args.strings_weight is not None else (4.0 if k <= 31 else 1.0)
    Aw = A / b[:, None] * wgt[:, None]
    # (fractional amounts: the generator makes floor(x) whole families and one more with the remaining fraction of the copies)
    frozen = {}
    if args.freeze_cores:
        old = json.load(open(os.path.join(ROOT, "sshash_amd", "recipes", args.name + ".json")))
        frozen = {(c["copies"], c["length"], c["core"]): c["families"] for c in old["classes"] if "core" in c and "note" not in c}
    fixed = np.zeros(A.shape[1])
    is_fixed = np.zeros(A.shape[1], dtype=bool)
    for i, mt in enumerate(meta):
        if "core" in mt:
            is_fixed[i] = bool(args.freeze_cores)
            fixed[i] = frozen.get((mt["copies"], mt["length"], mt["core"]), 0.0)
    if is_fixed.any():
        xf, rnorm = nnls(Aw[:, ~is_fixed], wgt - Aw[:, is_fixed] @ fixed[is_fixed], maxiter=50000)
        x = fixed.copy()
        x[~is_fixed] = xf
    else:
        x, rnorm = nnls(Aw, wgt, maxiter=50000)
    # classes the fit left at zero drop out; whatever else is small was rounded above
    fit = A @ x
    print(f"NNLS residual {rnorm:.4f}; classes in use: {int((x > 0).sum())} of {len(x)}", file=sys.stderr)
    for n, tt, ff in zip(names, b, fit):
        print(f"  {n:18s} aim {tt:14.0f} (published {tv[n]:14.0f}) fit {ff:14.0f} ({ff / tt - 1:+.1%})", file=sys.stderr)
    recipe_classes, bg = [], []
    for xi, mt in zip(x, meta):
        if xi <= 0:
            continue
        if "background_mean_len" in mt:
            bg.append({"mean_len": mt["background_mean_len"], "bases": xi * 1e6})
        else:
            cls = {kk: mt[kk] for kk in ("copies", "length", "divergence", "core") if kk in mt}
            recipe_classes.append(dict(cls, families=xi, max_bucket_alone=mt["max_bucket"]))
    tune = TAIL_TUNING.get(args.name, {})
    for c in recipe_classes:
        sd = tune.get("class_seeds", {}).get((c["copies"], c.get("core", 0)))
        if sd is not None:
            c["seed"] = sd
        c["families"] = max(0.0, c["families"] + tune.get("families_delta", {}).get((c["copies"], c.get("core", 0)), 0.0))
    recipe_classes += tune.get("extra_classes", [])
    recipe = {"name": args.name, "k": k, "m": m, "target": t, "reference_bases": t["num_bases"], "classes": recipe_classes,
              "background": bg, "fit": {n: float(f) for n, f in zip(names, fit)}, "aim": {n: float(aim[n]) for n in names}}
    os.makedirs(os.path.join(ROOT, "sshash_amd", "recipes"), exist_ok=True)
    path = os.path.join(ROOT, "sshash_amd", "recipes", args.name + ".json")
    json.dump(recipe, open(path, "w"), indent=1)
    print("wrote", path, file=sys.stderr)


if __name__ == "__main__":
    main()
