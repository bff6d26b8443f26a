#!/usr/bin/env python3
"""ISA audit of the asm-issued tile prefetch of k_decim_mfma (kernels_decim_mfma.hip tile_issue / tile_wait).

tile_issue loads 16-byte pairs into ACCUMULATOR registers with an asm statement the compiler does not count in its vmcnt
bookkeeping; tile_wait is the matching explicit `s_waitcnt vmcnt(0)`.  The technique is only sound if NO instruction touches one
of those registers between a load and the wait that covers it (a read would see stale data, a compiler copy would lose the
load).  This script proves that on the generated code: for every k_decim_mfma instantiation it follows every control-flow path
from each `global_load_dwordx4 a[..]` to the first `s_waitcnt vmcnt(0)` and reports any instruction on the way that names one of
the registers in flight.  Exit status 0 = clean.  Usage: audit_mfma_prefetch.py [file.s]   (without an argument the device
assembly is generated with hipcc -S)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "qradiolink_amd", "csrc")


def device_asm():
    out = os.path.join(tempfile.mkdtemp(prefix="qrl_audit_"), "mfma.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950",
                           "--cuda-device-only", "-S", "-o", out, "-I" + CSRC, os.path.join(CSRC, "kernels_decim_mfma.hip")],
                          stderr=subprocess.DEVNULL)
    return out


def regs(text):
    """set of AGPR indices named in an operand string"""
    r = set()
    for lo, hi in re.findall(r"\ba\[(\d+):(\d+)\]", text):
        r.update(range(int(lo), int(hi) + 1))
    for n in re.findall(r"\ba(\d+)\b", text):
        r.add(int(n))
    return r


def audit_function(name, lines):
    ins, labels = [], {}
    for ln in lines:
        t = ln.split(";")[0].strip()
        if not t or t.startswith("."):
            m = re.match(r"^(\.LBB\w+):", ln.strip())
            if m:
                labels[m.group(1)] = len(ins)
            continue
        m = re.match(r"^(\.LBB\w+):", t)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        ins.append(t)
    loads = [i for i, t in enumerate(ins) if t.startswith("global_load_dwordx4") and regs(t.split(",")[0])]
    problems, checked = [], 0
    for i0 in loads:
        mine = regs(ins[i0].split(",")[0])
        seen, stack = set(), [i0 + 1]
        while stack:
            i = stack.pop()
            while i < len(ins) and i not in seen:
                seen.add(i)
                t = ins[i]
                if t.startswith("s_waitcnt") and "vmcnt(0)" in t:
                    break
                if t.startswith("s_endpgm"):
                    break   # (path-insensitive walk: the wave ends with the load in flight into a dead register; harmless)
                hit = regs(t) & mine
                if hit:
                    if True:
                        problems.append("%s: '%s' touches a%s while the load '%s' is in flight" % (name, t, sorted(hit), ins[i0]))
                m = re.match(r"^(s_cbranch_\w+|s_branch)\s+(\.LBB\w+)", t)
                if m:
                    tgt = labels.get(m.group(2))
                    if tgt is not None:
                        stack.append(tgt)
                    if m.group(1) == "s_branch":
                        break
                i += 1
            checked += 1
    return len(loads), problems


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else device_asm()
    text = open(path).read().splitlines()
    funcs, cur, name = {}, None, None
    for ln in text:
        m = re.match(r"^(_ZN3qrl12k_decim_mfma\w+):", ln)
        if m:
            name, cur = m.group(1), []
            funcs[name] = cur
            continue
        if cur is not None:
            cur.append(ln)
            if "s_endpgm" in ln:
                cur = None
    total, bad = 0, []
    for name, lines in funcs.items():
        n, problems = audit_function(name, lines)
        total += n
        bad.extend(problems)
        print("%-90s %3d asm prefetch loads  %s" % (name[:90], n, "CLEAN" if not problems else "%d PROBLEMS" % len(problems)))
    for p in bad:
        print("  !!", p)
    print("%d k_decim_mfma instantiations, %d asm-issued loads audited, %d problems" % (len(funcs), total, len(bad)))
    return 1 if bad or not funcs or not total else 0


if __name__ == "__main__":
    sys.exit(main())
