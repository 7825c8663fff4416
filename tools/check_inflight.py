#!/usr/bin/env python3
"""check_inflight.py -- static check of the scan kernels' hand-issued asynchronous loads.

cds_scan.hip issues global loads through `asm volatile` and waits with hand-written `s_waitcnt vmcnt(N)`; the compiler
does not know the destination registers are pending, so a register copy, spill or any other read it might insert between
the load and the wait would silently read stale data.  This script walks the ISA of every scan_kernel instantiation
(`hipcc -S --cuda-device-only`) in layout order: a register written by an asm-issued global_load stays "pending" until the
next asm s_waitcnt; any instruction that reads or writes a pending register in between is reported.  Loop back-edges are
followed once (registers pending at the latch are checked against the loop body from its header).

    python tools/check_inflight.py            # compiles colormipsearch_amd/csrc/cds_scan.hip, exits 1 on a violation
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "colormipsearch_amd", "csrc", "cds_scan.hip")


def regs_of(tok):
    """v12 -> {12}; v[10:13] -> {10..13}"""
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


SEQ = [0]  # issue order of the asm loads (loads retire in order)


def walk(lines, labels, start, end, pending, name, problems, follow_back):
    """Walk lines[start:end] in layout order with the set of pending registers; on a backward branch re-walk the loop
    body once with the registers pending at the latch (that is where back-edge copies would read them)."""
    in_asm = False
    ln = start
    while ln < end:
        t = lines[ln].strip()
        ln += 1
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        t = t.split(";")[0].strip()
        if not t or t.endswith(":"):
            continue
        if in_asm and t.startswith("global_load"):
            ops = t.split(None, 1)[1].split(",")
            for r in regs_of(",".join(ops[1:])):
                if r in pending:
                    problems.append((name, ln, t, r))
            SEQ[0] += 1
            for r in regs_of(ops[0]):
                pending[r] = SEQ[0]
            continue
        if in_asm and t.startswith("s_waitcnt") and "vmcnt" in t:  # only the hand-written waits count (strict)
            n = int(re.search(r"vmcnt\((\d+)\)", t).group(1))
            if n == 0:
                pending.clear()
            else:  # loads retire in order: the youngest n loads may still be outstanding
                groups = sorted(set(pending.values()))
                keep = set(groups[-n:])
                for r in [r for r, g in pending.items() if g not in keep]:
                    del pending[r]
            continue
        if in_asm:
            continue
        for r in regs_of(t):
            if r in pending:
                problems.append((name, ln, t, r))
        m = re.match(r"s_c?branch\w*\s+(\.LBB\w+)", t)
        if m and follow_back and m.group(1) in labels and labels[m.group(1)] < ln - 1 and pending:
            walk(lines, labels, labels[m.group(1)], ln - 1, dict(pending), name, problems, False)


def check(asm_text):
    problems, kernels = [], 0
    for km in re.finditer(r"^(_ZN3cds11scan_kernel\w+):.*?\.end_amdhsa_kernel", asm_text, re.S | re.M):
        kernels += 1
        lines = km.group(0).splitlines()
        labels = {l.split(":")[0].strip(): i for i, l in enumerate(lines) if re.match(r"^\.LBB\w+:", l)}
        walk(lines, labels, 0, len(lines), {}, km.group(1), problems, True)
    problems = sorted(set(problems))
    return kernels, problems


def main():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "scan.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-device-only",
                        "-S", SRC, "-o", out], check=True, stderr=subprocess.DEVNULL)
        kernels, problems = check(open(out).read())
    print(f"{kernels} scan_kernel instantiations checked, {len(problems)} accesses to registers with a load in flight")
    for name, ln, t, r in problems[:40]:
        print(f"  {name} +{ln}: v{r} pending in: {t}")
    return 1 if problems or kernels == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
