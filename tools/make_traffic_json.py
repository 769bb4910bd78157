#!/usr/bin/env python
"""profiles/traffic.json from a PMC profile of the bench command (tools/jobs/r06_profile.sh -> <dir>/summary.json + bench.jsonl):
HBM bytes per bench step, with the counters, the formula and the commit it was measured at.

    python tools/make_traffic_json.py <profile dir> <kept summary path> [record name]

profiles/traffic.json holds one record per record name -- bench.py's `traffic_key`: the workload (c3, c2, c4) for the lookup line,
<workload>_streaming_p<percent positive> for the streaming lines, <workload>_directory / <workload>_mphf for the table-less paths
(SSHASH_AMD_SKTABLE=0 [SSHASH_AMD_DIRECTORY=0]) -- and bench.py attaches a record to a line when the shape it was measured on
(queries or reads per GPU, read length, bases, k, canonical) is the line's own. Every kernel of the step counts: the lookup passes
(first, resume, deferred, scan), the streaming kernels (pack pass, run kernel)."""
import json
import subprocess
import sys

src, kept = sys.argv[1], sys.argv[2]
s = json.load(open(src + "/summary.json"))
bench = json.loads(open(src + "/bench.jsonl").read().strip().splitlines()[-1])
cfg, roof = bench["config"], bench["roofline"]
streaming = "reads_per_gpu" in cfg
pl, ks = s["per_launch_counter_averages"], s["kernel_stats"]


def bytes_of(d):
    return d["TCC_EA0_RDREQ_sum"] * 64 + d["TCC_EA0_WRREQ_64B_sum"] * 64 + (d["TCC_EA0_WRREQ_sum"] - d["TCC_EA0_WRREQ_64B_sum"]) * 32


per_kernel = {k: bytes_of(pl[k]) for k in pl if "TCC_EA0_RDREQ_sum" in pl[k]}
launches = 1 if streaming else roof.get("launches_per_step", 1)  # a lookup step of 10^9 queries is eight launch sequences; their kernels are averaged per launch
total = int(sum(per_kernel.values()) * launches)
units = cfg["reads_per_gpu"] * (cfg["read_length"] - cfg["k"] + 1) if streaming else cfg["queries_per_gpu"]
requests = sum(pl[k]["TCC_EA0_RDREQ_sum"] + pl[k]["TCC_EA0_WRREQ_sum"] for k in per_kernel) * launches
name = sys.argv[3] if len(sys.argv) > 3 else bench.get("traffic_key")
if not name:
    raise SystemExit("no record name: the bench line carries no traffic_key and none was given")
rec = {
    "workload": cfg.get("workload_short") or cfg["workload"],
    "bases": cfg.get("num_bases"), "canonical": cfg["canonical"], "k": cfg["k"],
    "hbm_bytes_per_launch": total,
    "hbm_requests_per_unit": round(requests / units, 4),
    "unit_note": "bytes per STEP = %d x sum over the step's kernels (%s) of [TCC_EA0_RDREQ_sum x 64 B + TCC_EA0_WRREQ_64B_sum x 64 B + (TCC_EA0_WRREQ_sum - "
                 "TCC_EA0_WRREQ_64B_sum) x 32 B], averages per launch of `rocprofv3 --pmc` passes of the bench command itself; TCC_EA0_RDREQ_32B_sum = 0 (every read "
                 "request is 64 bytes); random 64-byte requests are counted once (calibrated with tools/tlb_probe: 2^27 random lines -> 1.342e8 RDREQ), so no x2 "
                 "correction for this access pattern; the coalesced streams (queries, ids, the reads' characters) may be under-counted by up to 2x (guide: wide "
                 "streaming reads tally 128-byte requests as 64)" % (launches, ", ".join(sorted(per_kernel))),
    "per_kernel_bytes": {k: int(v) for k, v in per_kernel.items()},
    "per_launch_counters": {k: {c: int(v) for c, v in pl[k].items() if c.startswith(("TCC", "TCP", "SQ_INSTS_VALU", "SQ_WAVES"))} for k in pl},
    "kernel_avg_ns": {k: ks[k]["avg_ns"] for k in ks},
    "counters": "rocprofv3 --pmc, one group per pass (tools/jobs/r06_profile.sh)",
    "commit": subprocess.check_output(["git", "rev-parse", "--short", "HEAD"]).decode().strip(),
    "source": kept,
}
if streaming:
    rec.update({"reads": cfg["reads_per_gpu"], "read_length": cfg["read_length"], "kmers": units})
else:
    rec.update({"queries": cfg["queries_per_gpu"], "hbm_requests_per_lookup": rec["hbm_requests_per_unit"]})
json.dump(s, open(kept, "w"), indent=1)
try:
    every = json.load(open("profiles/traffic.json"))
    if "queries" in every:  # (round-3 file: one record, C3's)
        every = {"c3": every}
except Exception:
    every = {}
every[name] = rec
json.dump(every, open("profiles/traffic.json", "w"), indent=1)
print(name, total / 1e9, "GB per step =", round(total / units, 2), "B per", "k-mer;" if streaming else "lookup;", rec["hbm_requests_per_unit"], "requests per unit")
