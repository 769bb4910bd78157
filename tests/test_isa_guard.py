"""CPU: tools/isa_guard.py (a 64-bit VALU instruction must not take a 32-bit operand from the LAST VGPR of its wave's allocation:
DESIGN.md section 6, tools/debug/vgpr64_check.hip) -- the rule on hand-made assembly, and the SHIPPED library: every kernel of
every gfx950 code object inside libsshash_amd.so, allocation from its kernel descriptor, instructions from its disassembly."""
from __future__ import annotations

import json
import os
import re
import shutil
import struct
import subprocess
import sys

import pytest

from conftest import ROOT

LLVM = "/opt/rocm/lib/llvm/bin"

KERNEL = """\t.text
\t.globl\t{name}
\t.type\t{name},@function
{name}:
\ts_load_dwordx2 s[0:1], s[0:1], 0x0
{body}
\ts_endpgm
\t.section\t.rodata,"a",@progbits
\t.amdhsa_kernel {name}
\t\t.amdhsa_group_segment_fixed_size 0
\t\t.amdhsa_next_free_vgpr {vgprs}
\t\t.amdhsa_next_free_sgpr 16
\t\t.amdhsa_accum_offset {accum}
\t.end_amdhsa_kernel
\t.text
.Lfunc_end_{name}:
"""


def guard(tmp_path, text, *flags):
    src, dst, rep = tmp_path / "in.s", tmp_path / "out.s", tmp_path / "report.json"
    src.write_text(text)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "isa_guard.py"), str(src), str(dst), "--report", str(rep)] + list(flags))
    return dst.read_text(), json.loads(rep.read_text())


CASES = {
    # name: (body, next_free_vgpr, exposed?)
    "shift_by_last": ("\tv_lshlrev_b64 v[60:61], v63, v[60:61]\n\tv_add_u32_e32 v5, v0, v62", 64, True),
    "right_shift_by_last_of_56": ("\tv_lshrrev_b64 v[2:3], v55, v[2:3]", 56, True),
    "mad_with_last_as_factor": ("\tv_mad_u64_u32 v[2:3], s[4:5], v5, v63, 0", 64, True),
    "indexed_registers": ("\tv_lshlrev_b64 v[60:61], v63, v[60:61]\n\tv_movrels_b32_e32 v1, v8", 64, True),
    "value_in_the_last_pair": ("\tv_lshlrev_b64 v[62:63], v5, v[62:63]", 64, False),
    "thirty_two_bit_shift_by_last": ("\tv_lshlrev_b32_e32 v1, v63, v2", 64, False),
    "last_register_not_used": ("\tv_lshlrev_b64 v[60:61], v61, v[60:61]\n\tv_mov_b32_e32 v61, 0", 62, False),
    "last_as_32_bit_destination": ("\tv_cvt_u32_f64_e32 v63, v[2:3]", 64, False),
    "not_the_last": ("\tv_lshlrev_b64 v[60:61], v63, v[60:61]\n\tv_mov_b32_e32 v70, 0", 71, False),
}


def test_the_rule_on_hand_made_assembly(tmp_path):
    text = "".join(KERNEL.format(name=n, body=b, vgprs=v, accum=(v + 3) // 4 * 4) for n, (b, v, _) in CASES.items())
    # padding only: an exposed kernel gets eight more registers, its code is not touched
    out, report = guard(tmp_path, text, "--pad-only")
    padded = {p["kernel"]: p for p in report["padded"]}
    assert report["kernels"] == len(CASES) and not report["renamed"]
    for name, (body, vgprs, hit) in CASES.items():
        assert (name in padded) == hit, name
        block = out[out.index(".amdhsa_kernel " + name):]
        now = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", block).group(1))
        assert now == ((vgprs + 7) // 8 * 8 + 8 if hit else vgprs), name
        assert body in out
    again, report2 = guard(tmp_path, out, "--pad-only")
    assert again == out and not report2["padded"]
    # by default: registers renamed (same allocation), padding only where names are not all there is (M0-relative indexing)
    out, report = guard(tmp_path, text)
    renamed = {p["kernel"]: p for p in report["renamed"]}
    assert set(renamed) == {"shift_by_last", "right_shift_by_last_of_56", "mad_with_last_as_factor"}
    assert [p["kernel"] for p in report["padded"]] == ["indexed_registers"]
    assert renamed["shift_by_last"]["registers"] == "v[62:63] <-> v[60:61]"
    # the same program under other names: the shifted pair and the amount have changed places, v0 (the workitem id) stays v0
    assert "v_lshlrev_b64 v[62:63], v61, v[62:63]" in out and "v_add_u32_e32 v5, v0, v60" in out
    for name in renamed:
        block = out[out.index(".amdhsa_kernel " + name):]
        assert int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", block).group(1)) == CASES[name][1]
    again, report2 = guard(tmp_path, out)
    assert again == out and not report2["padded"] and not report2["renamed"]


def code_objects(tmp_path, lib):
    work = tmp_path / ("unbundle_" + os.path.basename(lib))
    work.mkdir()
    shutil.copy(lib, work / "lib.so")
    subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=work, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return sorted(str(p) for p in work.iterdir() if "gfx950" in p.name and p.stat().st_size > 0)


def exposed_kernels(tmp_path, lib):
    """-> (kernels examined, [(code object, kernel, allocation, instruction)]) over every gfx950 code object inside `lib`: allocation from the
    kernel descriptor, instructions from the disassembly -- the FINAL binary, whatever built it"""
    objs = code_objects(tmp_path, lib)
    assert len(objs) >= 3  # engine, streaming, sktable
    wide = re.compile(r"^v_\w*(b64|u64|i64|f64)")
    checked, exposed = 0, []
    for obj in objs:
        syms = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "-s", "-S", "-W", obj], text=True)
        rodata = next(l.split() for l in syms.splitlines() if re.search(r"\]\s+\.rodata\s", l))
        i = rodata.index(".rodata")
        sec_addr, sec_off = int(rodata[i + 2], 16), int(rodata[i + 3], 16)
        blob = open(obj, "rb").read()
        alloc = {}
        for l in syms.splitlines():
            f = l.split()
            if len(f) >= 8 and f[-1].endswith(".kd"):
                kd = blob[sec_off + int(f[1], 16) - sec_addr:][:64]
                rsrc1 = struct.unpack_from("<I", kd, 48)[0]
                alloc[f[-1][:-3]] = ((rsrc1 & 0x3F) + 1) * 8
        assert alloc
        name = None
        for l in subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", obj], text=True).splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", l)
            if m:
                name = m.group(1)
                checked += name in alloc
                continue
            if name not in alloc or not l.startswith("\t"):
                continue
            code = l.split("//")[0].strip()
            parts = code.split(None, 1)
            if len(parts) < 2 or not wide.match(parts[0]):
                continue
            last = alloc[name] - 1
            single = re.compile(r"(?<![\w\[:])v%d\b(?!\s*:)" % last)
            if any(single.search(o) for o in parts[1].split(",")[1:]):
                exposed.append((os.path.basename(obj), name, alloc[name], code))
    return checked, exposed


def test_no_kernel_of_the_shipped_library_is_exposed(tmp_path):
    lib = os.environ.get("SSHASH_TEST_LIBRARY") or os.path.join(ROOT, "sshash_amd", "libsshash_amd.so")  # (another build: to see the test fail)
    if not os.path.exists(lib):
        pytest.skip("library not built")
    checked, exposed = exposed_kernels(tmp_path, lib)
    assert checked > 100
    assert not exposed, exposed[:5]


def test_the_plain_build_shows_what_the_guard_is_for(tmp_path):
    """`make PLAIN=1` (hipcc end to end, no guard; built by __graft_entry__.build() next to the shipped library, never shipped): the same
    examination NAMES the kernels this toolchain's register allocation exposes. A toolchain that exposes none makes the guard idle, one
    that exposes others shows here first -- a red or changed CPU test, not wrong answers on the GPU. The two builds hold the same kernels."""
    plain = os.path.join(ROOT, "sshash_amd", "csrc", "build", "plain", "libsshash_amd_plain.so")
    shipped = os.path.join(ROOT, "sshash_amd", "libsshash_amd.so")
    if not (os.path.exists(plain) and os.path.exists(shipped)):
        pytest.skip("make -C sshash_amd/csrc PLAIN=1 has not been run")
    if os.path.getmtime(plain) + 1 < max(os.path.getmtime(os.path.join(ROOT, "sshash_amd", "csrc", f)) for f in ("engine.hip", "streaming.hip", "sktable.hip", "lookup_device.hpp")):
        pytest.skip("the plain build is older than the sources")
    checked_plain, exposed = exposed_kernels(tmp_path, plain)
    checked_shipped, none = exposed_kernels(tmp_path, shipped)
    assert checked_plain == checked_shipped and not none
    names = sorted({k for _, k, _, _ in exposed})
    print("kernels of the PLAIN build exposed to the last-VGPR hazard:", len(names))
    for n in names:
        print("  ", n)
    # with hipcc 7.2 (ROCm 7.2.0) these are instances of fast_lookup_kernel / resume_lookup_kernel in engine.hip; the guard's own report of
    # the shipped build names the same kernels as "renamed" or "padded"
    reports = [json.load(open(os.path.join(ROOT, "sshash_amd", "csrc", "build", f + ".isa_guard.json"))) for f in ("engine", "streaming", "sktable")]
    repaired = sorted({r["kernel"] for rep in reports for r in rep["renamed"] + rep["padded"]})
    assert names == repaired
    assert all(rep["toolchain"] and "HIP version" in rep["toolchain"] for rep in reports)
    assert sum(rep["kernels"] for rep in reports) == checked_shipped  # every kernel of the library went through the guard


REFUSED = {
    # what the guard must not let through unexamined: name -> (body, the descriptor's register count, text expected in the refusal)
    "register_count_is_an_expression": ("\tv_mov_b32_e32 v1, v2", "max(64, callee.num_vgpr)", "not a plain integer"),
    "calls_a_function": ("\ts_getpc_b64 s[4:5]\n\ts_swappc_b64 s[30:31], s[4:5]", 64, "calls a function"),
    "jumps_through_a_register": ("\ts_setpc_b64 s[30:31]", 64, "calls a function"),
    "accumulation_registers": ("\tv_accvgpr_write_b32 a0, v1", 64, "accumulation registers"),
    "unreadable_line": ("\t%%% what is this", 64, "not an instruction"),
}


@pytest.mark.parametrize("name", sorted(REFUSED))
def test_the_guard_fails_closed(name, tmp_path):
    body, vgprs, why = REFUSED[name]
    text = KERNEL.format(name="fine", body="\tv_mov_b32_e32 v1, v2", vgprs=8, accum=8) + KERNEL.format(name=name, body=body, vgprs=vgprs, accum=64)
    src, dst = tmp_path / "in.s", tmp_path / "out.s"
    src.write_text(text)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_guard.py"), str(src), str(dst)], capture_output=True, text=True)
    assert p.returncode == 2 and "REFUSED" in p.stderr and why in p.stderr and name in p.stderr, p.stderr
    assert not dst.exists()


def test_a_long_branch_is_not_a_call(tmp_path):
    body = "\ts_getpc_b64 s[98:99]\n.Lpost_getpc1:\n\ts_add_u32 s98, s98, 16\n\ts_addc_u32 s99, s99, 0\n\ts_setpc_b64 s[98:99]"
    out, report = guard(tmp_path, KERNEL.format(name="long_branch", body=body, vgprs=8, accum=8))
    assert report["kernels"] == 1 and body in out


def test_the_library_says_how_it_was_built():
    from sshash_amd import _binding

    if not os.path.exists(_binding.library_path()):
        pytest.skip("library not built")
    assert _binding.build_info() == {"isa_guard": "guarded", "arch": "gfx950"}
