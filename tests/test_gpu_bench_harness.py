"""GPU: bench.py as the driver runs it -- `python bench.py --gpus N --steps K --warmup W` -- on a reduced workload:
one JSON line on stdout with the contract's fields, the roofline and cpu_baseline objects at N = 1, and the N = 2 harness
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


def run_bench(extra, env_extra, tmp_path):
    env = dict(os.environ, SSHASH_BENCH_CACHE=str(tmp_path), **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + extra, env=env, capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, f"stdout must carry exactly one line, got {len(lines)}: {p.stdout[:500]}"
    return json.loads(lines[0])


def test_one_gpu_line_carries_the_contract(tmp_path):
    r = run_bench([], {}, tmp_path)
    for key in CONTRACT:
        assert key in r, key
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
    assert set(r["other_paths"]) == {"directory", "mphf"} and all(v["ids_equal_table_path"] and v["lookups_per_s"] > 0 for v in r["other_paths"].values())
    f = r["streaming_from_file"]
    assert f["counters_equal_oracle_on_sample"] is True and f["kmers"] == 200000 * 120
    for flavour in ("fastq", "fastq.gz", "bgzf.fastq.gz"):
        rep = f[flavour]["report"]
        assert rep["num_kmers"] == f["kmers"] == rep["num_positive_kmers"] + rep["num_negative_kmers"] + rep["num_invalid_kmers"]
        assert f[flavour]["ns_per_kmer"] > 0 and f[flavour]["reader_alone"]["reads"] == 200000
    assert f["fastq"]["report"] == f["fastq.gz"]["report"] == f["bgzf.fastq.gz"]["report"]


def test_two_ranks_split_one_batch(tmp_path):
    r = run_bench(["--gpus", "2", "--no-cpu-baseline"], {"SSHASH_BENCH_TEST_ALL_RANKS_ON_DEVICE": "0"}, tmp_path)
    assert r["n_gpus"] == 2 and r["scaling"] == "strong"
    assert [p["rank"] for p in r["per_rank"]] == [0, 1]
    assert sum(p["queries"] for p in r["per_rank"]) == r["config"]["queries_per_step"] == 4000000
    assert r["ms_per_step"] >= max(p["ms_per_step"] for p in r["per_rank"]) * 0.98  # the MAX over the ranks
    assert len([f for f in os.listdir(tmp_path) if f.endswith(".sshash")]) == 1  # built once, by rank 0
    assert r["cpu_baseline"] is None and r["config"]["index_replicated_per_gpu"] is True


@pytest.mark.parametrize("how", ["table", "minimizer"])
def test_two_ranks_route_one_batch_over_a_partitioned_dictionary(how, tmp_path):
    """`bench.py --sharded ... --gpus 2`: the N > 1 ROUTED path (route -> all-to-all -> lookup -> return -> combine), both ranks on
    the one GPU of the test box and the exchange over gloo; ids are checked against the oracle inside bench.py."""
    r = run_bench(["--gpus", "2", "--no-cpu-baseline", "--sharded", how], {"SSHASH_BENCH_TEST_ALL_RANKS_ON_DEVICE": "0"}, tmp_path)
    assert r["n_gpus"] == 2 and r["config"]["sharded"] == how and r["config"]["index_replicated_per_gpu"] is False
    assert sum(p["queries"] for p in r["per_rank"]) == 4000000 and abs(r["config"]["positive_fraction_found"] - 0.5) < 0.01


def test_config_c4_line(tmp_path):
    """`bench.py --workload c4` (human k = 63, m = 25 stand-in) at reduced size: the two-word path behind the same line -- statistics
    against the published k = 63 build, the table-less paths, and the FASTQ query with its counters equal to the oracle's."""
    env = dict(os.environ, SSHASH_BENCH_CACHE=str(tmp_path))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    args = ["--workload", "c4", "--bases", "30000000", "--queries", "2000000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--file-reads", "100000"]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, p.stderr[-3000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.strip()][-1])
    assert r["config"]["k"] == 63 and r["config"]["m"] == 25 and r["dtype"] == "u64" and r["config"]["recipe"] == "human_k63"
    assert r["config"]["index_statistics"]["source"].startswith("benchmarks/results-10-11-25/k63/regular-build.log")
    assert abs(r["config"]["positive_fraction_found"] - 0.5) < 0.01
    assert all(v["ids_equal_table_path"] for v in r["other_paths"].values())
    f = r["streaming_from_file"]
    assert f["counters_equal_oracle_on_sample"] is True and f["kmers"] == 100000 * (150 - 63 + 1)
    assert f["fastq"]["report"] == f["fastq.gz"]["report"] and f["fastq"]["report"]["num_positive_kmers"] > 0
    assert f["published_reference"]["ns_per_kmer"] == 190.6


def test_streaming_mode_one_rank_and_two(tmp_path):
    """`bench.py --workload c4 --streaming [--gpus 2]` (BASELINE.json configs[3]) at reduced size: reads drawn on every rank's device,
    read-sharded, the six counters summed with one all_reduce; the line carries the contract, per-rank reports and the oracle check."""
    env = dict(os.environ, SSHASH_BENCH_CACHE=str(tmp_path))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    base = ["--workload", "c4", "--bases", "30000000", "--streaming", "--reads", "400001", "--steps", "2", "--warmup", "1", "--stream-oracle-reads", "5000"]
    lines = {}
    for n, extra, env_extra in ((1, [], {}), (2, ["--gpus", "2"], {"SSHASH_BENCH_TEST_ALL_RANKS_ON_DEVICE": "0"})):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + base + extra, env=dict(env, **env_extra), capture_output=True, text=True, timeout=1200)
        assert p.returncode == 0, p.stderr[-3000:]
        out = [l for l in p.stdout.splitlines() if l.strip()]
        assert len(out) == 1
        lines[n] = r = json.loads(out[0])
        for key in CONTRACT:
            assert key in r, key
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
    env = dict(os.environ, SSHASH_BENCH_CACHE=str(tmp_path), SSHASH_BENCH_TEST_OTHER_WORKLOADS="24000000,2000000,200000")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    args = ["--bases", "24000000", "--queries", "2000000", "--steps", "2", "--warmup", "1", "--cpu-sample", "100000", "--no-extra-mixes", "--no-other-paths", "--no-file-query"]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=2400)
    assert p.returncode == 0, p.stderr[-3000:]
    out = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(out) == 1
    r = json.loads(out[0])
    others = r["other_workloads"]
    assert set(others) == {"c2", "c4", "c4_streaming", "c3_streaming_high_hit", "c3_streaming_high_hit_table_key_25"}
    assert others["c3_streaming_high_hit_table_key_25"]["environment"] == {"SSHASH_AMD_SK_M": "25"} and "environment" not in others["c3_streaming_high_hit"]
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
