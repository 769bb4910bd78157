#!/usr/bin/env python
"""profiles/traffic.json from a PMC profile of the bench command (tools/jobs/r02_profile.sh -> <dir>/summary.json):
HBM bytes per bench step, with the counters, the formula and the commit it was measured at.

    python tools/make_traffic_json.py gpurun_out/r04_prof_c3 profiles/r04/bench_c3_pmc_summary.json [bases] [workload name]

profiles/traffic.json holds one record per workload name (c3, c2, c4: bench.py --workload); bench.py attaches the record of the
workload it runs when the shape (queries, bases, k, canonical) is the same."""
import json
import subprocess
import sys

src, kept = sys.argv[1], sys.argv[2]
s = json.load(open(src + "/summary.json"))
bench = json.loads(open(src + "/bench.jsonl").read().strip().splitlines()[-1])
if len(sys.argv) <= 3:
    sys.argv.append(str(bench["config"].get("num_bases", 2_813_192_630)))
workload = sys.argv[4] if len(sys.argv) > 4 else "c3"
pl, ks = s["per_launch_counter_averages"], s["kernel_stats"]
launches = bench["roofline"]["launches_per_step"]


def bytes_of(d):
    return d["TCC_EA0_RDREQ_sum"] * 64 + d["TCC_EA0_WRREQ_64B_sum"] * 64 + (d["TCC_EA0_WRREQ_sum"] - d["TCC_EA0_WRREQ_64B_sum"]) * 32


per_seq = {k: bytes_of(pl[k]) for k in ("fast", "resume", "deferred") if k in pl and "TCC_EA0_RDREQ_sum" in pl[k]}  # (k <= 31: no resume pass since round 3)
total = int(sum(per_seq.values()) * launches)
n = bench["config"]["queries_per_gpu"]
requests = sum(pl[k]["TCC_EA0_RDREQ_sum"] + pl[k]["TCC_EA0_WRREQ_sum"] for k in per_seq) * launches
rec = {
    "workload": bench["config"]["workload"],
    "queries": n, "bases": int(sys.argv[3]) if len(sys.argv) > 3 else 2_813_192_630,  # bench.py --bases (default: the C3 workload's)
    "canonical": bench["config"]["canonical"], "k": bench["config"]["k"],
    "hbm_bytes_per_launch": total,
    "hbm_requests_per_lookup": round(requests / n, 4),
    "unit_note": "bytes per STEP (%d launch sequences) = %d x sum over the lookup kernels of one launch sequence (first pass, deferred pass; a resume pass at k > 31) of [TCC_EA0_RDREQ_sum x 64 B + "
                 "TCC_EA0_WRREQ_64B_sum x 64 B + (TCC_EA0_WRREQ_sum - TCC_EA0_WRREQ_64B_sum) x 32 B], averages per launch; TCC_EA0_RDREQ_32B_sum = 0 "
                 "(every read request is 64 bytes); random 64-byte requests are counted once (calibrated with tools/tlb_probe: 2^27 random lines -> "
                 "1.342e8 RDREQ), so no x2 correction for this access pattern; the coalesced query/id streams (16 B per lookup) may be under-counted "
                 "by up to 2x (guide: wide streaming reads tally 128-byte requests as 64)" % (launches, launches),
    "per_launch_sequence_bytes": {k: int(v) for k, v in per_seq.items()},
    "per_launch_counters": {k: {c: int(v) for c, v in pl[k].items() if c.startswith(("TCC", "TCP"))} for k in pl},
    "kernel_avg_ns": {k: ks[k]["avg_ns"] for k in ks},
    "counters": "rocprofv3 --pmc, one group per pass (tools/jobs/r03_profile.sh)",
    "commit": subprocess.check_output(["git", "rev-parse", "--short", "HEAD"]).decode().strip(),
    "source": kept,
}
json.dump(s, open(kept, "w"), indent=1)
try:
    every = json.load(open("profiles/traffic.json"))
    if "queries" in every:  # (round-3 file: one record, C3's)
        every = {"c3": every}
except Exception:
    every = {}
every[workload] = rec
json.dump(every, open("profiles/traffic.json", "w"), indent=1)
print(total / 1e9, "GB per step =", round(total / n, 2), "B per lookup;", rec["hbm_requests_per_lookup"], "requests per lookup")
