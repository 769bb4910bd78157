"""GPU: the environment switches that change what a replica holds or how a batch is launched (SSHASH_AMD_SKTABLE,
_DIRECTORY, _SK_M, _SK_DENSITY, and the tests' own SSHASH_AMD_TEST_HOOKS), each in a process of its own (tests/gpu_switch_worker.py): every k-mer of
a C2-like stand-in through the id-returning and the is_member instances, launch after launch -- 20 launches where the round-3 hazard
lived --, ASCII input, the bench's mixes against the oracle. (Round 5 removed the A/B-only switches -- INWAVE, OVERLAP -- and their kernels.)"""
from __future__ import annotations

import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

SETTINGS = [
    ("default", {}, 20),
    ("packed_table", {"SSHASH_AMD_TEST_HOOKS": "slots_per_key=1.4,slots_per_kmer=1.3"}, 3),
    ("density_compact", {"SSHASH_AMD_SK_DENSITY": "compact"}, 3),   # the documented footprint switch (round 6): 1.6 slots per item instead of 2.5
    ("density_number", {"SSHASH_AMD_SK_DENSITY": "2.0", "SSHASH_AMD_SK_SLOTS_PER_KEY": "9"}, 3),  # (the removed knob is named on stderr and not read)
    ("small_pieces", {"SSHASH_AMD_TEST_HOOKS": "piece=1000000"}, 3),
    ("table_key_17", {"SSHASH_AMD_SK_M": "17"}, 3),   # the table's own key length (sk_view::m): shorter and longer than the
    ("table_key_25", {"SSHASH_AMD_SK_M": "25"}, 3),   # dictionary's minimizers (m = 21 here)
    ("directory", {"SSHASH_AMD_SKTABLE": "0", "SSHASH_AMD_DIRECTORY": "1"}, 3),
    ("mphf", {"SSHASH_AMD_SKTABLE": "0", "SSHASH_AMD_DIRECTORY": "0"}, 3),
]
WORKLOADS = {"c2_like_regular": ("se_k31", 20_000_000, 31, 21, 0), "c3_like_canonical": ("human_k31", 12_000_000, 31, 21, 1),
             "c4_like_k63": ("human_k63", 12_000_000, 63, 25, 0)}


def run(workload, env, launches):
    recipe, bases, k, m, canonical = WORKLOADS[workload]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "gpu_switch_worker.py"), recipe, str(bases), str(k), str(m), str(canonical), str(launches)],
                       capture_output=True, text=True, timeout=1500, env=dict(os.environ, **env))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("name,env,launches", SETTINGS, ids=[s[0] for s in SETTINGS])
def test_every_kmer_launch_after_launch(name, env, launches):
    got = run("c2_like_regular", env, launches)
    assert got["ok"] and got["launches"] == launches
    if name in ("directory", "mphf"):
        assert got["sk_slots"] == 0 and (name == "directory") == bool(got["directory_sectors"])
    else:
        assert got["sk_slots"] > 0
    if name.startswith("density"):  # fewer slots than the default replica of the same dictionary, the same answers
        assert got["sk_slots"] < run("c2_like_regular", {}, 1)["sk_slots"] * (0.7 if name == "density_compact" else 0.85)


@pytest.mark.parametrize("workload", ["c3_like_canonical", "c4_like_k63"])
def test_other_flavours_launch_after_launch(workload):
    assert run(workload, {}, 20)["ok"]
