#!/usr/bin/env python3
"""Writes r/patches/chicdiff_hip.patch: the hunks a maintainer applies to Chicdiff/R/chicdiff.R so that the reference's own
functions reach the MI355X path.  Eleven added lines, two changed; every reference statement stays where
it is — the device path enters getRegionUniverse / getFullRegionData at their first line, DESeq2Wrap AFTER the reference's
own argument handling and warnings (chicdiff.R:1496-1521), IHWcorrection at its two covariate lines (:1965, :1979) and around
its application block (:2038-2049).  Needs the reference tree to cut the hunks from (run where /root/reference exists);
tests/test_r_shim.py checks that the committed patch still applies to it."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/Chicdiff/R/chicdiff.R"
HIP = 'if (identical(chicdiff.settings[["backend"]], "hip"))'


def main():
    src = open(REF).read()
    L = src.split("\n")

    def find(prefix, start=0):
        for i in range(start, len(L)):
            if L[i].startswith(prefix):
                return i
        raise SystemExit("not found: " + prefix)

    new, ins = list(L), []
    i = find('getRegionUniverse <- function(chicdiff.settings, suffix = ""){')
    ins.append((i, [f"  {HIP} return(.getRegionUniverseHip(chicdiff.settings, suffix = suffix)) ## MI355X path"]))
    i = find('getFullRegionData <- function(chicdiff.settings, RU, RUcontrol, suffix = ""){')
    ins.append((i, [f"  {HIP} return(.getFullRegionDataHip(chicdiff.settings, RU, RUcontrol, suffix = suffix)) ## MI355X path"]))
    i0 = find('DESeq2Wrap <- function(chicdiff.settings, RU, FullRegionData, suffix = "", theta = NULL){')
    i = find("  ##Input data:", i0)
    ins.append((i - 1, [f"  {HIP} ## MI355X path: arguments, theta and norm as resolved above",
                        "    return(.DESeq2WrapHip(chicdiff.settings, RU, FullRegionData, suffix = suffix, theta = theta, norm = norm))", ""]))
    i1 = find("IHWcorrection <- function(chicdiff.settings, DESeqOut, FullRegionData, DESeqOutControl, FullControlRegionData,")
    for var, tab in (("RU.distances", "RU.recast"), ("RU.distancesControl", "RU.recastControl")):
        a = find(f"  {var} <- {tab}[", i1)
        rhs = L[a].split("<-", 1)[1].strip()
        new[a] = f"  {var} <- if (.isHipRegionData({tab})) .hipRegionDistances({tab}) else {rhs}"
    c0 = find("  out[,avgLogDist := log(abs(avDist))]", i1)
    c1 = find("  out[, weighted_padj := p.adjust(weighted_pvalue", i1)
    ins.append((c0 - 1, [f"  {HIP} {{ ## MI355X path: cut, weight look-up, weighted p and BH on the device",
                         "    out <- .hipApplyIHWweights(out, distLookup, device = .hipDeviceIndex(chicdiff.settings))", "  } else {"]))
    ins.append((c1, ["  }"]))
    for idx, lines in sorted(ins, key=lambda t: -t[0]):
        new[idx + 1:idx + 1] = lines
    with tempfile.TemporaryDirectory() as td:
        for side, text in (("a", src), ("b", "\n".join(new))):
            os.makedirs(os.path.join(td, side, "Chicdiff", "R"))
            open(os.path.join(td, side, "Chicdiff", "R", "chicdiff.R"), "w").write(text)
        p = subprocess.run(["diff", "-U1", "--label", "a/Chicdiff/R/chicdiff.R", "--label", "b/Chicdiff/R/chicdiff.R",
                            "a/Chicdiff/R/chicdiff.R", "b/Chicdiff/R/chicdiff.R"], cwd=td, capture_output=True, text=True)
    out = os.path.join(ROOT, "r", "patches", "chicdiff_hip.patch")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    open(out, "w").write(p.stdout)
    plus = sum(1 for l in p.stdout.split("\n") if l.startswith("+") and not l.startswith("+++"))
    minus = sum(1 for l in p.stdout.split("\n") if l.startswith("-") and not l.startswith("---"))
    print(f"{out}: {plus} lines added, {minus} removed")


if __name__ == "__main__":
    main()
