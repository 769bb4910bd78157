#!/usr/bin/env python
"""RESULTS.md's table from the full record of the driver's command (bench.py --full-record, default bench_full.json):

    python tools/make_results_md.py profiles/r06/bench_driver_command_full.json [driver's BENCH_rNN.json]

One row per configuration: the number, its time, roofline fraction on algorithmic bytes, PMC traffic (bytes per unit and fraction of the
HBM peak), the 64-byte requests against this box's random-line probe, the CPU restatement on the same host. Printed to stdout."""
import json
import sys

full = json.load(open(sys.argv[1]))
driver = json.load(open(sys.argv[2])) if len(sys.argv) > 2 else None


def g(v, digits=2):
    return "—" if v is None else f"{v / 1e9:.{digits}f}"


def row(name, what, w):
    roof, cpu = w.get("roofline") or {}, w.get("cpu_baseline") or {}
    unit = "lookups" if w["unit"].startswith("lookups") else "k-mers"
    per = roof.get("hbm_traffic_bytes_per_lookup", roof.get("hbm_traffic_bytes_per_kmer"))
    alg = roof.get("algorithmic_bytes_per_lookup", roof.get("algorithmic_bytes_per_kmer"))
    bound = (roof.get("random_unit_bound") or {})
    box = (bound.get("this_box") or {})
    return (f"| {name} | {what} | **{g(w['value'])}** G {unit}/s | {w['ms_per_step']} | {roof.get('frac')} ({alg} B) | "
            f"{'—' if per is None else per} B = {roof.get('frac_hbm_traffic', '—')} | {box.get('frac', bound.get('frac', '—'))} | "
            f"{cpu.get('value', 0) / 1e6:.2f} M/s on {cpu.get('cores')} thread(s) |")


print("| config | workload | value | ms per step | frac of 8 TB/s on algorithmic bytes (bytes per unit) | PMC traffic per unit = frac of peak | 64-byte requests ÷ this box's random-line probe | CPU restatement, same host |")
print("|---|---|---|---|---|---|---|---|")
cfg = full["config"]
print(row("C3 (headline)", f"k={cfg['k']} m={cfg['m']}, {cfg['num_kmers'] / 1e9:.2f} G k-mers, {cfg['queries_per_step'] / 1e9:.0f} G queries/step, {cfg['device_bytes_per_kmer']} B/k-mer in HBM", full))
names = {"c2": "C2", "c4": "C4 (k = 63)", "c4_streaming": "C4 streaming_query", "c3_streaming_high_hit": "k = 31 streaming_query, high-hit"}
for key, w in (full.get("other_workloads") or {}).items():
    if "error" in w:
        print(f"| {names.get(key, key)} | {w['error']} | | | | | | |")
        continue
    c = w["config"]
    what = (f"k={c['k']} m={c['m']}, {c['num_kmers'] / 1e9:.2f} G k-mers, " +
            (f"{c['reads'] / 1e6:.0f} M reads x {c['read_length']}, {c['positive_fraction_of_kmers']:.0%} of the k-mers positive" if "reads" in c
             else f"{c['queries_per_step'] / 1e9:.1f} G queries/step, {c['device_bytes_per_kmer']} B/k-mer in HBM"))
    print(row(names.get(key, key), what, w))
for name, v in (full.get("other_paths") or {}).items():
    if name.startswith("host_"):  # the host-buffer entry points: PCIe inclusive, wall clock of the call
        what = "sshash_lookup_packed" if name == "host_packed" else "sshash_lookup_ascii (k-mers as characters)"
        print(f"| C3, host arrays (PCIe inclusive): {what} | {v['queries'] / 1e6:.0f} M queries of the headline batch in page-locked, device-mapped caller arrays; ids equal the device path's | "
              f"**{g(v['lookups_per_s'])}** G lookups/s | {v['ms']} | | {v['link_GBps_both_directions']} GB/s over the link, both directions together | — | — |")
        continue
    print(f"| C3 without the table: {name} | 10^8 queries of the headline batch, ids equal the table path's | **{g(v['lookups_per_s'])}** G lookups/s | {v['ms']} | {v['roofline_frac']} | "
          f"{v.get('hbm_traffic_bytes_per_lookup') or '—'} B = {v.get('frac_hbm_traffic') or '—'} | — | — |")
for name, v in (full.get("other_mixes") or {}).items():
    print(f"| C3, other mix: {name} | 10^8 queries, {v['fraction_found']:.0%} found | **{g(v['lookups_per_s'])}** G lookups/s | {v['ms']} | | | | |")
f = full.get("streaming_from_file") or {}
for fl in ("fastq", "fastq.gz", "bgzf.fastq.gz"):
    if fl in f:
        print(f"| `sshash query` end to end, {fl} | {f['reads'] / 1e6:.0f} M reads x {f['read_length']} | **{g(f[fl]['kmers_per_s'])}** G k-mers/s = {f[fl]['ns_per_kmer']} ns per k-mer | | | | | oracle: {f['cpu_oracle']['ns_per_kmer']} ns per k-mer |")
probe = ((full["roofline"].get("random_unit_bound") or {}).get("this_box") or {})
print()
print(f"Box: random-line probe {g(probe.get('probe_units_per_s'), 1)} G lines/s (tools/tlb_probe); commit of the traffic records: "
      f"{(full['roofline'].get('traffic_provenance') or {}).get('commit')}.")
if driver:
    p = driver.get("parsed") or {}
    print(f"Driver's own record: value {g(p.get('value'))} G lookups/s, {p.get('ms_per_step')} ms per step, roofline frac {(p.get('roofline') or {}).get('frac')}.")
