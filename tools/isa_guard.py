#!/usr/bin/env python3
"""isa_guard.py -- keep a gfx950 hardware hazard out of the shipped kernels (tools/debug/vgpr64_check.hip).

The hazard, as measured on MI355X (profiles/r04/vgpr64_check_*.jsonl): a VALU instruction of a 64-bit data type whose 32-BIT source
operand is the LAST VGPR of the wave's allocation -- v63 of 64, v55 of 56, v71 of 72 (allocations are granules of 8) -- returns a
wrong result in 6-7 % of its executions, differently from launch to launch: v_lshlrev_b64 / v_lshrrev_b64 / v_ashrrev_i64 with the
shift amount there, for one. The same instruction is right with the operand in any other register, with the 64-bit VALUE in the
last pair, with the wave given eight more registers, and after a v_mov of the operand into another register. hipcc 7.2 knows no such
constraint (the operand is 32 bits wide: any VGPR will do), so whether a kernel is hit is an accident of register allocation: the
is_member instance of fast_lookup_kernel allocates exactly 64 registers and keeps a shift amount in v63.

Which instructions: of those tried (tools/debug/vgpr64_check.hip modes 0-18, profiles/r04/vgpr64_check_*.jsonl) the three 64-bit
shifts fail -- v_lshlrev_b64, v_lshrrev_b64, v_ashrrev_i64 --; v_mad_u64_u32 (either factor, or the addend pair), v_lshl_add_u64,
v_cvt_f64_u32, 32-bit shifts and multiplies by that register and LDS addresses in it are right. The guard stays on the safe side of
what was tried: ANY 64-bit-typed VALU instruction with a 32-bit source in the last register.

What this does: reads the device assembly of one translation unit (hipcc --cuda-device-only -S), finds every kernel that (1) uses
the last register of its allocation and (2) names it as a 32-bit operand of an instruction whose mnemonic carries a 64-bit type, and
repairs it without touching what the kernel computes:
  rename  the aligned block of 2, 4 or 8 registers that ends the allocation swaps NAMES with another block, throughout the kernel --
          the same program (the experiment that found the hazard: profiles/r04/member_race_3_*.txt) --, chosen so that every register
          tuple of the kernel stays contiguous and aligned, the workitem id stays in v0, and the register that ends up last is not
          such an operand. Nothing is lost: same allocation, same occupancy;
  pad     where no such block exists (or the kernel indexes registers through M0): eight more registers (.amdhsa_next_free_vgpr,
          .amdhsa_accum_offset, .vgpr_count) -- the register named is no longer the last one; one wave per SIMD less may fit.
Writes the assembly back out and a JSON report; csrc/Makefile assembles what comes out.

The guard FAILS CLOSED: it raises -- and the build stops -- when it meets something it was not written for, instead of
letting a kernel through unexamined: a kernel descriptor whose register count is not a plain integer (hipcc writes `max(...)`
expressions for kernels that call non-inlined functions), a kernel that calls (s_swappc / s_call, or an s_setpc that is not the
assembler's own long branch: the callee's registers are not the kernel's names), a number of kernels examined that differs from the number of `.amdhsa_kernel` directives in the file, or a
line inside a kernel's body that is neither an instruction, a label, a directive nor a comment. The report records the compiler
that wrote the assembly (`--toolchain`).

    python3 tools/isa_guard.py in.s out.s [--report report.json] [--check] [--pad-only] [--toolchain "text"]     (--check: exit 1 if anything had to be changed)
"""
import json
import re
import sys

GRANULE = 8
# 64-bit typed VALU mnemonics: ..._b64, _u64, _i64, _f64 anywhere in the name (v_lshlrev_b64, v_mad_u64_u32, v_lshl_add_u64, v_cvt_f64_u32 ...)
WIDE = re.compile(r"^v_\w*(b64|u64|i64|f64)")


class GuardError(Exception):
    """something in the assembly the guard cannot vouch for: the build must stop"""


INSTRUCTION = re.compile(r"^[a-z][a-z0-9_]*(\s|$)")  # a mnemonic: v_..., s_..., ds_..., global_..., buffer_..., flat_..., scratch_...


def kernels_of(lines):
    """-> [{name, body: (first, last) line indices, vgpr_line, accum_line, next_free_vgpr}]"""
    labels = {}
    for i, l in enumerate(lines):
        if l and not l[0].isspace() and l.rstrip().split(";")[0].rstrip().endswith(":"):
            labels.setdefault(l.split(":")[0].strip(), i)
    out = []
    i = 0
    while i < len(lines):
        m = re.match(r"\s*\.amdhsa_kernel\s+(\S+)", lines[i])
        if m:
            k = {"name": m.group(1), "desc": i}
            j = i
            while ".end_amdhsa_kernel" not in lines[j]:
                mm = re.match(r"\s*\.amdhsa_next_free_vgpr\s+(.*?)\s*(;.*)?$", lines[j])
                if mm:
                    if not re.fullmatch(r"\d+", mm.group(1)):
                        raise GuardError("kernel %s: .amdhsa_next_free_vgpr is not a plain integer (%r): a kernel with non-inlined calls? "
                                         "The guard cannot tell its last register" % (k["name"], mm.group(1)))
                    k["vgpr_line"], k["next_free_vgpr"] = j, int(mm.group(1))
                mm = re.match(r"\s*\.amdhsa_accum_offset\s+(\d+)", lines[j])
                if mm:
                    k["accum_line"], k["accum_offset"] = j, int(mm.group(1))
                j += 1
            if k["name"] not in labels:
                raise GuardError("kernel %s: no label of that name in the file" % k["name"])
            if "next_free_vgpr" not in k:
                raise GuardError("kernel %s: its descriptor has no .amdhsa_next_free_vgpr" % k["name"])
            k["body"] = (labels[k["name"]], i)
            out.append(k)
            i = j
        i += 1
    directives = sum(1 for l in lines if re.match(r"\s*\.amdhsa_kernel\s", l))
    if len(out) != directives:
        raise GuardError("%d .amdhsa_kernel directives but %d kernels parsed" % (directives, len(out)))
    return out


def body_end(lines, k):
    """the line after a kernel's last instruction: its .Lfunc_end label (the descriptor follows in another section)"""
    for i in range(k["body"][0], k["body"][1]):
        if re.match(r"\.Lfunc_end\d*\w*:", lines[i]) or re.match(r"\s*\.section\s", lines[i]):
            return i
    return k["body"][1]


def classify_body(lines, k):
    """every line of the kernel's body must be something the guard understands; a kernel that calls is refused"""
    for i in range(k["body"][0], body_end(lines, k)):
        l = lines[i]
        code = l.split(";")[0].strip()
        if not code or code.startswith(".") or code.endswith(":"):
            continue  # empty / comment, directive, label
        if code.split("//")[0].strip() == "":
            continue
        if not INSTRUCTION.match(code):
            raise GuardError("kernel %s, line %d: not an instruction, label, directive or comment: %r" % (k["name"], i + 1, l))
        mnemonic = code.split(None, 1)[0]
        if re.search(r"(?<![\w.])a(\d+|\[\d+:\d+\])(?![\w(])", code.split(None, 1)[1] if " " in code or "\t" in code else ""):
            raise GuardError("kernel %s, line %d: accumulation registers in use (%s): with them the allocation is split at .amdhsa_accum_offset "
                             "and 'the last VGPR' is not what this guard computes" % (k["name"], i + 1, code))
        if mnemonic.startswith("s_setpc"):
            # a long branch inside the kernel (s_getpc_b64 sN, label arithmetic, s_setpc_b64 sN: what the assembler's branch relaxation
            # writes when a target is out of a 16-bit offset's reach) is not a call; anything else that sets the PC is refused
            pair = code.split(None, 1)[1].strip()
            recent = [lines[j].split(";")[0].strip() for j in range(max(k["body"][0], i - 6), i)]
            if any(r.startswith("s_getpc_b64") and r.split(None, 1)[1].strip() == pair for r in recent):
                continue
        if mnemonic.startswith(("s_swappc", "s_setpc", "s_call")):
            raise GuardError("kernel %s, line %d: %s -- the kernel calls a function that was not inlined; its registers follow the "
                             "callee's convention, not the names in this body" % (k["name"], i + 1, mnemonic))


def hazards(lines, k):
    n = k["next_free_vgpr"]
    alloc = (max(n, 1) + GRANULE - 1) // GRANULE * GRANULE
    last = alloc - 1
    if n - 1 < last:
        return alloc, []  # the last register of the allocation is not used at all
    single = re.compile(r"(?<![\w\[:])v%d\b(?!\s*:)" % last)  # v63 as an operand of its own, not inside v[62:63]
    found = []
    for i in range(k["body"][0], k["body"][1]):
        code = lines[i].split(";")[0].strip()
        if not code or code.startswith(".") or code.endswith(":"):
            continue
        parts = code.split(None, 1)
        if len(parts) < 2 or not WIDE.match(parts[0]):
            continue
        operands = parts[1].split(",")
        # sources only: the first operand is the destination (a 32-bit destination in the last register is fine, and rare)
        if any(single.search(o) for o in operands[1:]):
            found.append({"line": i + 1, "instruction": code})
    return alloc, found


TUPLE = re.compile(r"\bv\[(\d+):(\d+)\]")
SINGLE = re.compile(r"\bv(\d+)\b")


def body_code(lines, k):
    """(index, code) of the instruction lines of a kernel's body"""
    for i in range(k["body"][0], k["body"][1]):
        l = lines[i]
        if l.lstrip().startswith(";;"):
            continue
        code = l.split(";")[0]
        c = code.strip()
        if c and not c.startswith(".") and not c.endswith(":"):
            yield i, code


def try_rename(lines, k, alloc):
    """-> (new lines of the body as {index: text}, description) or None"""
    last = alloc - 1
    code = list(body_code(lines, k))
    text = "\n".join(c for _, c in code)
    if re.search(r"\bv_movrel|\bs_set_gpr_idx", text):
        return None  # registers addressed through M0: names are not all there is
    tuples = {(int(a), int(b)) for a, b in TUPLE.findall(text)}
    ids = 1
    for i in range(k["desc"], k["desc"] + 64):
        m = re.match(r"\s*\.amdhsa_system_vgpr_workitem_id\s+(\d+)", lines[i])
        if m:
            ids = int(m.group(1)) + 1
        if ".end_amdhsa_kernel" in lines[i]:
            break

    def block_ok(base, size):
        return all(b < base or a >= base + size or (a >= base and b < base + size) for a, b in tuples)

    for size in (2, 4, 8):
        high = last // size * size
        if not block_ok(high, size):
            continue
        for low in range(high - size, -1, -size):
            if low < ids or not block_ok(low, size):
                continue
            delta = high - low

            def swap(n):
                return n - delta if high <= n < high + size else n + delta if low <= n < low + size else n

            def rename(c):
                c = TUPLE.sub(lambda m: "v[%d:%d]" % (swap(int(m.group(1))), swap(int(m.group(2)))), c)
                return SINGLE.sub(lambda m: "v%d" % swap(int(m.group(1))), c)

            new = {i: rename(c) for i, c in code}
            trial = list(lines)
            for i, c in new.items():
                trial[i] = c
            if not hazards(trial, k)[1]:
                return new, "v[%d:%d] <-> v[%d:%d]" % (high, high + size - 1, low, low + size - 1)
    return None


def main():
    argv = sys.argv[1:]
    args, skip = [], False
    for a in argv:
        if skip:
            skip = False
        elif a in ("--report", "--toolchain"):
            skip = True
        elif not a.startswith("--"):
            args.append(a)
    src, dst = args[0], args[1]
    report_path = sys.argv[sys.argv.index("--report") + 1] if "--report" in sys.argv else None
    lines = open(src).read().split("\n")
    toolchain = sys.argv[sys.argv.index("--toolchain") + 1] if "--toolchain" in sys.argv else None
    report = {"source": src, "toolchain": toolchain, "kernels": 0, "use_their_last_register": 0, "renamed": [], "padded": []}
    bumped = {}
    for k in kernels_of(lines):
        report["kernels"] += 1
        classify_body(lines, k)
        alloc, found = hazards(lines, k)
        if k["next_free_vgpr"] == alloc:
            report["use_their_last_register"] += 1
        if not found:
            continue
        renamed = None if "--pad-only" in sys.argv else try_rename(lines, k, alloc)
        if renamed:
            for i, c in renamed[0].items():
                lines[i] = c
            report["renamed"].append({"kernel": k["name"], "allocation": alloc, "registers": renamed[1], "instructions": found[:8], "count": len(found)})
            continue
        new = alloc + GRANULE
        lines[k["vgpr_line"]] = re.sub(r"\d+\s*$", str(new), lines[k["vgpr_line"]])
        if "accum_line" in k:
            lines[k["accum_line"]] = re.sub(r"\d+\s*$", str(new), lines[k["accum_line"]])
        bumped[k["name"]] = new
        report["padded"].append({"kernel": k["name"], "allocation": alloc, "now": new, "instructions": found[:8], "count": len(found)})
    # the metadata copy of the register count (what hipFuncGetAttributes reports)
    name = None
    for i, l in enumerate(lines):
        m = re.match(r"\s*(?:- )?\.name:\s+(\S+)", l)
        if m:
            name = m.group(1)
        m = re.match(r"(\s*\.vgpr_count:\s+)(\d+)", l)
        if m and name in bumped:
            lines[i] = m.group(1) + str(bumped[name])
    open(dst, "w").write("\n".join(lines))
    if report_path:
        json.dump(report, open(report_path, "w"), indent=1)
    print("[isa_guard] %s: %d kernels, %d use the last register of their allocation, %d renamed, %d padded%s%s" % (
        src, report["kernels"], report["use_their_last_register"], len(report["renamed"]), len(report["padded"]),
        "".join("\n  " + p["kernel"][:100] + " %s: %s" % (p["registers"], p["instructions"][0]["instruction"]) for p in report["renamed"]),
        "".join("\n  " + p["kernel"][:100] + " %d -> %d: %s" % (p["allocation"], p["now"], p["instructions"][0]["instruction"]) for p in report["padded"])), file=sys.stderr)
    if "--check" in sys.argv and (report["padded"] or report["renamed"]):
        sys.exit(1)


if __name__ == "__main__":
    try:
        main()
    except GuardError as e:
        print("[isa_guard] REFUSED: %s" % e, file=sys.stderr)
        sys.exit(2)
