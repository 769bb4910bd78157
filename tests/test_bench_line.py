"""CPU: bench.py's stdout line, made from a full record of the driver's command kept under profiles/ -- round 4's line had grown to 34 KB
and the driver could not parse it. The compact form carries the contract's keys, `roofline` and `cpu_baseline`, one short entry per side
measurement, and stays below the limit also at eight ranks."""
from __future__ import annotations

import copy
import json
import os

from conftest import ROOT

import bench

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "cpu_baseline")


def full_record():
    return json.load(open(os.path.join(ROOT, "profiles", "r05", "bench_driver_command_full.json")))


def test_the_line_of_the_drivers_command_is_compact_and_complete():
    full = full_record()
    assert len(json.dumps(full)) > 20000  # (what the line used to be)
    line = bench.compact_line(full, "bench_full.json")
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) < bench.STDOUT_LINE_LIMIT // 2
    for key in CONTRACT:
        assert key in line, key
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert line["roofline"][key] == full["roofline"][key]
    assert line["roofline"]["traffic"] and 0 < line["roofline"]["frac"] < 1
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in line["cpu_baseline"]
    assert "workload" in line["config"] and "model" not in line["config"] and "device_stats" not in line["config"]
    assert set(line["other_workloads"]) == {"c2", "c4", "c4_streaming", "c3_streaming_high_hit"}
    for name, short in line["other_workloads"].items():
        child = full["other_workloads"][name]
        assert short["value"] == child["value"] and short["roofline_frac"] == child["roofline"]["frac"] and short["parity"]
        # (lookups: algorithmic bytes are what the kernel has to move at least -- below the roofline. Streaming: the reference's own bytes, which
        # the kernel is free to skip -- reported, labelled `frac_is`, not asserted to stay below 1: ADVICE r5)
        assert 0 < short["roofline_frac"] and (short["roofline_frac"] < 1 or "streaming" in name), (name, short)
        assert short["traffic"], name                            # PMC traffic on every line that has a kernel
    assert json.loads(text) == line


def test_the_line_stays_compact_at_eight_ranks():
    full = copy.deepcopy(full_record())
    full["n_gpus"] = 8
    full["per_rank"] = [dict(full["per_rank"][0], rank=r, upload_s=2.1, upload_window_s=[0.0, 2.1], random_line_probe_units_per_s=4.4e10) for r in range(8)]
    full["cpu_baseline"] = None
    full["cpu_baseline_note"] = "the CPU path is timed by the N = 1 run only"
    for key in ("other_mixes", "other_paths", "streaming_from_file", "other_workloads"):
        full[key] = None
    line = bench.compact_line(full, "bench_full.json")
    assert len(json.dumps(line, separators=(",", ":"))) < bench.STDOUT_LINE_LIMIT // 2
    assert len(line["per_rank"]) == 8 and line["cpu_baseline"] is None and "N = 1" in line["cpu_baseline_note"]
