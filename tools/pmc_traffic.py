#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE — they do not fit one pass, MI355X_MICROARCH.md
"rocprofv3 PMC slots") of `bench.py` into per-kernel HBM bytes per launch.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch --output-format csv -- \
        python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-kernels
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write --output-format csv -- \
        python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-kernels
    python3 tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles r01_e 2000000 8

Units and gfx950 corrections (guide, HBM/rocprofv3 section): both counters are in KiB; WRITE_SIZE is
exact for streaming stores; FETCH_SIZE under-counts streaming reads on gfx950 (exactly 1/2 for wide
16 B/lane loads).  The fit kernels load 4 B/lane (counts) and 8 B/lane (offsets), so the factor is
calibrated in the same run on kernels whose byte count is known: `row_lgm_kernel` (4 B/lane, 4·S·n bytes)
and `prep16_kernel` (4 + 8 B/lane, 12·S·n bytes — the same mix the fit kernels read).
"""
import glob
import json
import os
import sys

import pandas as pd


def per_kernel(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    df = pd.concat([pd.read_csv(f) for f in files])
    df = df[df["Counter_Name"] == counter]
    df["k"] = df["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace(r"^void (cd::)?", "", regex=True).str.replace("cd::", "")
    # the first (warm-up) launches are included: same work every step
    g = df.groupby(["k", "Counter_Name"])["Counter_Value"].agg(["mean", "count"]).reset_index()
    return g


def main():
    fetch_dir, write_dir, outdir, tag, n, S = sys.argv[1:7]
    n, S = int(n), int(S)
    f = per_kernel(fetch_dir, "FETCH_SIZE")
    w = per_kernel(write_dir, "WRITE_SIZE")
    f.to_csv(os.path.join(outdir, f"{tag}_pmc_fetch_size_bench_{n // 1000000}Mx{S}.csv"), index=False)
    w.to_csv(os.path.join(outdir, f"{tag}_pmc_write_size_bench_{n // 1000000}Mx{S}.csv"), index=False)
    fm = {r.k: r["mean"] for _, r in f.iterrows()}
    wm = {r.k: r["mean"] for _, r in w.iterrows()}

    def find(m, prefix):
        ks = [k for k in m if k.startswith(prefix)]
        return m[ks[0]] if ks else None

    cal = {}
    lg = find(fm, "row_lgm_kernel")
    if lg:
        cal["row_lgm_kernel (4 B/lane)"] = 4.0 * S * n / (lg * 1024)
    pk = find(fm, "prep16_kernel") or find(fm, "prep_kernel")
    if pk:
        cal["prep kernel (4+8 B/lane)"] = 12.0 * S * n / (pk * 1024)
    ok = find(wm, "offsets16_kernel") or find(wm, "offsets_kernel")
    if ok:
        cal["WRITE_SIZE on offsets kernel (8*S*n bytes known)"] = 8.0 * S * n / (ok * 1024)
    corr = cal.get("prep kernel (4+8 B/lane)", 2.0)
    out = {}
    # (a line-search stage = its launch + the fitDispGrid launch behind it, round 6: both inside the scope bench.py times)
    for name, prefix, extra in (("disp_gene", "disp_fit_kernel<false", "disp_grid_kernel<false"), ("disp_map", "disp_fit_kernel<true", "disp_grid_kernel<true"),
                                ("wald_irls", "wald_irls_kernel", None)):
        fk, wk = find(fm, prefix), find(wm, prefix)
        if fk is None or wk is None:
            continue
        if extra and find(fm, extra) is not None and find(wm, extra) is not None:
            fk, wk = fk + find(fm, extra), wk + find(wm, extra)
        out[f"{name}:{n}x{S}"] = {
            "hbm_bytes_per_launch": int(fk * 1024 * corr + wk * 1024),
            "fetch_size_kib_raw": fk, "write_size_kib": wk, "fetch_correction": round(corr, 3),
            "calibration": {k: round(v, 3) for k, v in cal.items()},
            "algorithmic_bytes_per_launch": (12 * S + 48) * n,
            "source": f"profiles/{tag}_pmc_fetch_size_*.csv, profiles/{tag}_pmc_write_size_*.csv (separate --pmc passes)",
        }
    json.dump(out, open(os.path.join(outdir, "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
