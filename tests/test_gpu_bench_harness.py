"""GPU: bench.py as the driver runs it -- `python bench.py --gpus N --steps K --warmup W` -- on a reduced workload:
ONE compact JSON line on stdout (< 8 KB: the driver's parser lost round 4's 34 KB line) with the contract's fields, the roofline and
cpu_baseline objects at N = 1; the full record (children, histograms, per-step times) in the file --full-record names; and the N = 2 harness
(self-launched ranks, index built once, the batch split, MAX-reduced time, per-rank times) with both ranks on the one GPU of
the test box."""
from __future__ import annotations

import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

SMALL = ["--workload", "c2", "--bases", "30000000", "--queries", "4000000", "--steps", "3", "--warmup", "1", "--cpu-sample", "100000", "--file-reads", "200000"]
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "cpu_baseline")


LINE_LIMIT = 8192


def run(args, env_extra, tmp_path, timeout=1200):
    """bench.py with `args`: (the stdout line, the full record). The line is the ONE line of stdout, below the limit, carries the
    contract, and every number it repeats equals the full record's; stderr's tail names the headline."""
    env = dict(os.environ, SSHASH_BENCH_CACHE=str(tmp_path), **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    record = os.path.join(str(tmp_path), "bench_full.json")
    if os.path.exists(record):
        os.remove(record)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args + ["--full-record", record], env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, f"stdout must carry exactly one line, got {len(lines)}: {p.stdout[:500]}"
    assert len(lines[0]) < LINE_LIMIT, f"the stdout line has {len(lines[0])} bytes"
    line, full = json.loads(lines[0]), json.load(open(record))
    for key in CONTRACT:
        assert key in line, key
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert line[key] == full[key], key
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in line["roofline"] and line["roofline"][key] == full["roofline"][key], key
    assert (line["cpu_baseline"] is None) == (full["cpu_baseline"] is None)
    if line["cpu_baseline"] is not None:
        for key in ("value", "unit", "cores", "kind", "sample"):
            assert key in line["cpu_baseline"], key
        assert line["cpu_baseline"]["value"] == full["cpu_baseline"]["value"]
    assert "workload" in line["config"] and "model" not in line["config"]
    for heavy in ("index_statistics", "table_histogram", "device_stats"):
        assert heavy not in line["config"], heavy
    assert "kernel_ms_steps" not in line["roofline"]
    tail = p.stderr[-600:]
    assert "[bench] headline" in tail and f"{full['ms_per_step']} ms/step" in tail, tail  # whatever tail a log keeps carries the number
    return line, full


def run_bench(extra, env_extra, tmp_path):
    return run(SMALL + extra, env_extra, tmp_path)


def test_one_gpu_line_carries_the_contract(tmp_path):
    line, r = run_bench([], {}, tmp_path)
    assert set(line["other_mixes"]) == set(r["other_mixes"]) and set(line["other_paths"]) == {"directory", "mphf", "host_packed", "host_ascii"}
    assert line["streaming_from_file"]["fastq"]["ns_per_kmer"] == r["streaming_from_file"]["fastq"]["ns_per_kmer"]
    assert r["n_gpus"] == 1 and r["steps"] == 3 and r["warmup"] == 1 and r["scaling"] == "strong" and r["higher_is_better"] is True
    assert r["vs_baseline"] is None and r["data"] == "synthetic" and r["dtype"] == "u64"
    assert abs(r["value"] - r["config"]["queries_per_step"] / r["ms_per_step"] * 1e3) / r["value"] < 0.02
    roof = r["roofline"]
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert 0 < roof["frac"] < 1 and roof["avg_kernel_ms"] <= r["ms_per_step"] * 1.05
    cpu = r["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0 and "sample" in cpu
    assert abs(r["config"]["positive_fraction_found"] - 0.5) < 0.01
    assert set(r["other_mixes"]) == {"positive100", "negative100_random", "mix50_mutated_negatives"}
    # round 3: the stand-in's statistics next to the published ones, the table-side histogram, the table-less paths, `sshash query` end to end
    stats = r["config"]["index_statistics"]
    assert stats["source"].startswith("benchmarks/results-10-11-25/k31/regular-build.log") and stats["num_kmers"]["achieved"] == r["config"]["num_kmers"]
    assert {"target", "achieved", "ratio"} <= set(stats["num_minimizer_positions_of_buckets_in_skew_index"])
    hist = r["config"]["table_histogram"]
    assert sum(hist["keys_by_occurrences"].values()) == r["config"]["device_stats"]["sk_keys"]
    assert sum(hist["super_kmers_by_occurrences_of_their_key"].values()) == hist["super_kmers"]
    paths = r["other_paths"]
    assert set(paths) == {"directory", "mphf", "host_packed", "host_ascii"} and all(v["lookups_per_s"] > 0 for v in paths.values())
    assert paths["directory"]["ids_equal_table_path"] and paths["mphf"]["ids_equal_table_path"]
    # round 6: the host-buffer entry points (PCIe inclusive, page-locked caller arrays), ids equal the device path's
    assert paths["host_packed"]["ids_equal_device_path"] and paths["host_ascii"]["ids_equal_device_path"] and line["other_paths"]["host_packed"]["link_GBps_both_directions"] > 0
    f = r["streaming_from_file"]
    assert f["counters_equal_oracle_on_sample"] is True and f["kmers"] == 200000 * 120
    for flavour in ("fastq", "fastq.gz", "bgzf.fastq.gz"):
        rep = f[flavour]["report"]
        assert rep["num_kmers"] == f["kmers"] == rep["num_positive_kmers"] + rep["num_negative_kmers"] + rep["num_invalid_kmers"]
        assert f[flavour]["ns_per_kmer"] > 0 and f[flavour]["reader_alone"]["reads"] == 200000
    assert f["fastq"]["report"] == f["fastq.gz"]["report"] == f["bgzf.fastq.gz"]["report"]


def test_two_ranks_split_one_batch(tmp_path):
    line, r = run_bench(["--gpus", "2"], {"SSHASH_BENCH_TEST_ALL_RANKS_ON_DEVICE": "0"}, tmp_path)
    assert r["n_gpus"] == 2 and r["scaling"] == "strong"
    assert [p["rank"] for p in r["per_rank"]] == [0, 1] == [p["rank"] for p in line["per_rank"]]
    # round 5: the CPU path is the N = 1 run's and the line says so; no side measurement at N > 1; every rank reports its own upload
    # (the windows overlap: the ranks do not queue behind one another) and the random-line probe of its own device
    assert r["cpu_baseline"] is None and "N = 1" in r["cpu_baseline_note"] == line["cpu_baseline_note"]
    assert r["other_mixes"] is None and r["other_paths"] is None and r["streaming_from_file"] is None and r["other_workloads"] is None
    (a0, a1), (b0, b1) = (p["upload_window_s"] for p in r["per_rank"])
    assert max(a0, b0) < min(a1, b1), r["per_rank"]
    assert all(5e9 < p["random_line_probe_units_per_s"] < 8e10 for p in line["per_rank"]), line["per_rank"]  # (here the two probes share one GPU)
    assert sum(p["queries"] for p in r["per_rank"]) == r["config"]["queries_per_step"] == 4000000
    assert r["ms_per_step"] >= max(p["ms_per_step"] for p in r["per_rank"]) * 0.98  # the MAX over the ranks
    assert len([f for f in os.listdir(tmp_path) if f.endswith(".sshash")]) == 1  # built once, by rank 0
    assert r["cpu_baseline"] is None and r["config"]["index_replicated_per_gpu"] is True


@pytest.mark.parametrize("how", ["table", "minimizer"])
def test_two_ranks_route_one_batch_over_a_partitioned_dictionary(how, tmp_path):
    """`bench.py --sharded ... --gpus 2`: the N > 1 ROUTED path (route -> all-to-all -> lookup -> return -> combine), both ranks on
    the one GPU of the test box and the exchange over gloo; ids are checked against the oracle inside bench.py."""
    _, r = run_bench(["--gpus", "2", "--no-cpu-baseline", "--sharded", how, "--no-line-probe"], {"SSHASH_BENCH_TEST_ALL_RANKS_ON_DEVICE": "0"}, tmp_path)
    assert r["n_gpus"] == 2 and r["config"]["sharded"] == how and r["config"]["index_replicated_per_gpu"] is False
    assert sum(p["queries"] for p in r["per_rank"]) == 4000000 and abs(r["config"]["positive_fraction_found"] - 0.5) < 0.01


@pytest.mark.parametrize("mode", ["lookup", "streaming", "sharded_table", "sharded_minimizer"])
def test_eight_ranks_rehearsal(mode, tmp_path):
    """The driver's `bench.py --gpus 8` before an 8-GPU node ever runs it (round 6): eight self-launched ranks on the ONE device of
    the test box, coordination over gloo -- rendezvous, per-rank seeds, the share arithmetic (4000003 queries / 400001 reads do not
    divide by eight), the MAX over the ranks, the gather of the per-rank numbers, and for the routed modes the group size of the
    all-to-all. What it cannot show is RCCL itself: with a GPU per rank the same code takes `nccl` (asserted on the line's `exchange`)."""
    env = {"SSHASH_BENCH_TEST_ALL_RANKS_ON_DEVICE": "0"}
    small = ["--workload", "c2", "--bases", "12000000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-line-probe", "--gpus", "8"]
    if mode == "streaming":
        line, r = run(small + ["--streaming", "--reads", "400001", "--stream-oracle-reads", "2000"], env, tmp_path)
        assert r["n_gpus"] == 8 and len(r["per_rank"]) == 8 and [p["rank"] for p in r["per_rank"]] == list(range(8))
        assert sum(p["reads"] for p in r["per_rank"]) == 400001 and max(p["reads"] for p in r["per_rank"]) - min(p["reads"] for p in r["per_rank"]) <= 1
        rep = r["config"]["report"]
        assert rep["num_kmers"] == 400001 * (150 - 31 + 1) == sum(p["report"][0] for p in r["per_rank"])
        assert [sum(p["report"][i] for p in r["per_rank"]) for i in range(6)] == list(rep.values())
        assert len({tuple(p["report"]) for p in r["per_rank"]}) == 8  # every rank drew its own reads (per-rank seeds)
    else:
        extra = {"lookup": [], "sharded_table": ["--sharded", "table"], "sharded_minimizer": ["--sharded", "minimizer"]}[mode]
        line, r = run(small + ["--queries", "4000003"] + extra, env, tmp_path)
        assert r["n_gpus"] == 8 and len(r["per_rank"]) == 8 and [p["rank"] for p in r["per_rank"]] == list(range(8))
        assert sum(p["queries"] for p in r["per_rank"]) == r["config"]["queries_per_step"] == 4000003
        assert max(p["queries"] for p in r["per_rank"]) - min(p["queries"] for p in r["per_rank"]) <= 1
        assert abs(r["config"]["positive_fraction_found"] - 0.5) < 0.01  # (ids are checked against the oracle inside bench.py)
        if mode == "lookup":
            assert r["config"]["index_replicated_per_gpu"] is True and r["config"]["exchange"] is None
        else:
            assert r["config"]["index_replicated_per_gpu"] is False and r["config"]["sharded"] == extra[1]
            assert r["config"]["exchange"].startswith("all_to_all_single over gloo") and "TEST scaffolding" in r["config"]["exchange"]
    assert r["scaling"] == "strong" and (mode == "streaming" or r["cpu_baseline"] is None)
    assert r["ms_per_step"] >= max(p["ms_per_step"] for p in r["per_rank"]) * 0.98  # the MAX over the ranks
    assert len([f for f in os.listdir(tmp_path) if f.endswith(".sshash")]) == 1  # built once, by rank 0


def test_config_c4_line(tmp_path):
    """`bench.py --workload c4` (human k = 63, m = 25 stand-in) at reduced size: the two-word path behind the same line -- statistics
    against the published k = 63 build, the table-less paths, and the FASTQ query with its counters equal to the oracle's."""
    args = ["--workload", "c4", "--bases", "30000000", "--queries", "2000000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--file-reads", "100000"]
    _, r = run(args, {}, tmp_path)
    assert r["config"]["k"] == 63 and r["config"]["m"] == 25 and r["dtype"] == "u64" and r["config"]["recipe"] == "human_k63"
    assert r["config"]["index_statistics"]["source"].startswith("benchmarks/results-10-11-25/k63/regular-build.log")
    assert abs(r["config"]["positive_fraction_found"] - 0.5) < 0.01
    assert all(v.get("ids_equal_table_path") or v.get("ids_equal_device_path") for v in r["other_paths"].values())
    f = r["streaming_from_file"]
    assert f["counters_equal_oracle_on_sample"] is True and f["kmers"] == 100000 * (150 - 63 + 1)
    assert f["fastq"]["report"] == f["fastq.gz"]["report"] and f["fastq"]["report"]["num_positive_kmers"] > 0
    assert f["published_reference"]["ns_per_kmer"] == 190.6


def test_streaming_mode_one_rank_and_two(tmp_path):
    """`bench.py --workload c4 --streaming [--gpus 2]` (BASELINE.json configs[3]) at reduced size: reads drawn on every rank's device,
    read-sharded, the six counters summed with one all_reduce; the line carries the contract, per-rank reports and the oracle check."""
    base = ["--workload", "c4", "--bases", "30000000", "--streaming", "--reads", "400001", "--steps", "2", "--warmup", "1", "--stream-oracle-reads", "5000"]
    for n, extra, env_extra in ((1, [], {}), (2, ["--gpus", "2"], {"SSHASH_BENCH_TEST_ALL_RANKS_ON_DEVICE": "0"})):
        line, r = run(base + extra, env_extra, tmp_path)
        assert line["config"]["report"] == r["config"]["report"] and line["roofline"]["traffic"] == r["roofline"]["traffic"]
        assert r["unit"] == "k-mers/s" and r["n_gpus"] == n and r["scaling"] == "strong" and r["config"]["k"] == 63
        rep = r["config"]["report"]
        assert rep["num_kmers"] == 400001 * (150 - 63 + 1) == sum(p_["report"][0] for p_ in r["per_rank"])
        assert [sum(p_["report"][i] for p_ in r["per_rank"]) for i in range(6)] == list(rep.values())
        assert sum(p_["reads"] for p_ in r["per_rank"]) == 400001 and len(r["per_rank"]) == n
        assert r["config"]["counters_equal_oracle_on_reads"] == 5000 and r["cpu_baseline"]["cores"] == 1
        assert 0.15 < r["config"]["positive_fraction_of_kmers"] < 0.5  # half of the reads spell k-mers of the dictionary, 1 % substitutions
        assert abs(r["value"] - rep["num_kmers"] / r["ms_per_step"] * 1e3) / r["value"] < 0.02
    assert len([f for f in os.listdir(tmp_path) if f.endswith(".sshash")]) == 1


def test_default_run_appends_the_other_baseline_configurations(tmp_path):
    """The driver's command measures C3 and, behind it, BASELINE.json's other single-GPU configurations as child runs of the same script
    (`other_workloads`: C2, C4, C4's streaming query, the k = 31 streaming query on high-hit reads), each with its own oracle check, roofline and cpu_baseline -- here at reduced size."""
    args = ["--bases", "24000000", "--queries", "2000000", "--steps", "2", "--warmup", "1", "--cpu-sample", "100000", "--no-extra-mixes", "--no-other-paths", "--no-file-query"]
    stdout_line, r = run(args, {"SSHASH_BENCH_TEST_OTHER_WORKLOADS": "24000000,2000000,200000"}, tmp_path, timeout=2400)
    others = r["other_workloads"]
    assert set(others) == {"c2", "c4", "c4_streaming", "c3_streaming_high_hit"} == set(stdout_line["other_workloads"])
    for name, short in stdout_line["other_workloads"].items():  # per child: value, unit, time, roofline fraction, CPU baseline, parity -- and nothing else
        assert set(short) == {"value", "unit", "ms_per_step", "steps", "roofline_frac", "frac_hbm_traffic", "traffic", "cpu_baseline_value", "cpu_cores", "parity"}
        assert short["value"] == others[name]["value"] and short["roofline_frac"] == others[name]["roofline"]["frac"]
        assert short["cpu_baseline_value"] == others[name]["cpu_baseline"]["value"] and short["parity"] in ("ids equal oracle", "counters equal oracle")
    for name, line in others.items():
        assert "error" not in line, line
        for key in CONTRACT:
            assert key in line, (name, key)
        assert line["value"] > 0 and line["roofline"]["frac"] > 0 and line["cpu_baseline"]["value"] > 0 and line["n_gpus"] == 1
    assert others["c2"]["config"]["k"] == 31 and others["c4"]["config"]["k"] == 63 and others["c4_streaming"]["unit"] == "k-mers/s"
    high = others["c3_streaming_high_hit"]
    assert high["unit"] == "k-mers/s" and high["config"]["k"] == 31 and high["config"]["positive_fraction_of_kmers"] > 0.5
    assert others["c2"]["config"]["recipe"] == "se_k31" and others["c4"]["config"]["recipe"] == "human_k63"
    # the random-line probe of the box the line was measured on (tools/tlb_probe), attached to every lookup line's random_unit_bound
    for line in (r, others["c2"], others["c4"]):
        box = line["roofline"]["random_unit_bound"]["this_box"]
        assert "error" not in box, box
        assert 2e10 < box["probe_units_per_s"] < 8e10 and box["frac"] > 0
