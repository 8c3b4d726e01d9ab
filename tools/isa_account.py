#!/usr/bin/env python3
"""ISA-level account of one line-search tick.

Compiles disp_kernels.hip for gfx950 with -DCHICDIFF_ISA_MARK — the MARK() statements of the kernel become comment lines in
the assembly — and counts, for disp_fit_kernel<MAP, MINW>, the instructions between consecutive marks in layout order, by
class: VALU (fp64 full rate / quarter rate: v_rcp_f64, conversions to or from f64, v_ldexp/frexp count as full rate), other
VALU (integer, moves, compares, permutes), SALU, LDS, VMEM, waitcnt.  Counts are STATIC (layout order): a loop body is counted once
— the trip counts that apply are given in the notes.  The marked build is for counting only (volatile asm pins the schedule).

usage: python tools/isa_account.py [--map] [--minw 2] > profiles/r04_isa_tick_account.txt
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QUARTER = re.compile(r"^v_(rcp|rsq|sqrt|div_scale|div_fmas|div_fixup)_f64|^v_cvt_\w*f64|^v_cvt_f64|^v_(ceil|floor|rndne|trunc|fract)_f64")


def classify(op):
    if op.startswith("v_"):
        if QUARTER.match(op):
            return "valu_f64_quarter"
        if "_f64" in op:
            return "valu_f64"
        return "valu_other"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    return "other"


def parts():
    """Per-part kernels (tools/ubench/tick_parts.hip): every part of a tick compiled alone, its instructions counted exactly."""
    src = os.path.join(ROOT, "tools", "ubench", "tick_parts.hip")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "parts.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-S",
                        "--cuda-device-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
        text = open(out).read().splitlines()
    cols = ["valu_f64", "valu_f64_quarter", "valu_other", "salu", "lds", "vmem", "waitcnt"]
    rows = {}
    i = 0
    while i < len(text):
        m = re.match(r"^(part_\w+):", text[i])
        if not m:
            i += 1
            continue
        cnt = collections.Counter()
        j = i + 1
        while ".end_amdhsa_kernel" not in text[j] and not text[j].startswith("\t.section"):
            t = text[j].strip()
            if t and not t.startswith((";", ".")) and not t.endswith(":"):
                cnt[classify(t.split()[0])] += 1
            j += 1
        rows[m.group(1)] = cnt
        i = j
    print("# every part of a line-search tick compiled ALONE (tools/ubench/tick_parts.hip -> gfx950 assembly): instructions of the part's kernel")
    print("# minus those of its frame (loads, stores, log-table set-up: part_frame; kernels without the table: part_rcp's frame ~ part_exp's).")
    print("# issue cycles = 4 per full-rate wave64 VALU instruction, 16 per quarter-rate one (v_rcp_f64, conversions, ceil).")
    print(f"{'part':28s}" + "".join(f"{c:>18s}" for c in cols) + f"{'VALU':>8s}{'issue cycles':>14s}")
    for name, cnt in rows.items():
        valu = cnt["valu_f64"] + cnt["valu_f64_quarter"] + cnt["valu_other"]
        cyc = 4 * (cnt["valu_f64"] + cnt["valu_other"]) + 16 * cnt["valu_f64_quarter"]
        print(f"{name:28s}" + "".join(f"{cnt[c]:18d}" for c in cols) + f"{valu:8d}{cyc:14d}")
    return rows


def main():
    if "--parts" in sys.argv:
        parts()
        return 0
    ap = argparse.ArgumentParser()
    ap.add_argument("--map", action="store_true")
    ap.add_argument("--minw", type=int, default=2)
    ap.add_argument("--extra", default="", help="extra compiler flags")
    args = ap.parse_args()
    src = os.path.join(ROOT, "chicdiff_amd", "csrc", "disp_kernels.hip")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "disp.s")
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-DCHICDIFF_ISA_MARK",
               "-S", "--cuda-device-only", src, "-o", out] + args.extra.split()
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        text = open(out).read().splitlines()
    sym = f"_ZN2cd15disp_fit_kernelILb{1 if args.map else 0}ELi{args.minw}EEEvNS_8DispArgsE"
    start = next(i for i, l in enumerate(text) if l.startswith(sym + ":"))
    end = next(i for i in range(start, len(text)) if ".end_amdhsa_kernel" in text[i])
    body = text[start:end]
    regs = [l.strip() for l in text[end:end + 60] if re.search(r"; (NumVgprs|NumSgprs|ScratchSize|Occupancy|codeLenInByte|LDSByteSize)", l)]
    sections, cur = [], ["(kernel prologue)", collections.Counter()]
    for l in body:
        t = l.strip()
        if t.startswith("; MARK "):
            sections.append(cur)
            cur = [t[7:], collections.Counter()]
            continue
        if not t or t.startswith((";", ".", "_Z")) or t.endswith(":"):
            continue
        cur[1][classify(t.split()[0])] += 1
    sections.append(cur)
    cols = ["valu_f64", "valu_f64_quarter", "valu_other", "salu", "lds", "vmem", "waitcnt"]
    print(f"# {'disp_fit_kernel<' + ('true' if args.map else 'false') + ', ' + str(args.minw) + '>'} — static instruction counts between marks, layout order")
    print("# " + "; ".join(regs[:6]))
    print(f"{'code up to mark':44s}" + "".join(f"{c:>18s}" for c in cols) + f"{'VALU total':>12s}{'issue cycles':>14s}")
    tot = collections.Counter()
    for name, cnt in sections:
        valu = cnt["valu_f64"] + cnt["valu_f64_quarter"] + cnt["valu_other"]
        cyc = 4 * (cnt["valu_f64"] + cnt["valu_other"]) + 16 * cnt["valu_f64_quarter"]
        print(f"{name:44s}" + "".join(f"{cnt[c]:18d}" for c in cols) + f"{valu:12d}{cyc:14d}")
        tot.update(cnt)
    valu = tot["valu_f64"] + tot["valu_f64_quarter"] + tot["valu_other"]
    print(f"{'whole kernel (static)':44s}" + "".join(f"{tot[c]:18d}" for c in cols) + f"{valu:12d}")


if __name__ == "__main__":
    sys.exit(main())
