"""Host logic of the pipeline mirrors (chicdiff_amd/settings.py, pipeline.py) and the guard that keeps the R host from
reinterpreting one of the reference's settings keys (`device` is the reference's PLOT device)."""
import os
import re

import numpy as np
import pytest

from pipeline_inputs import REPLICATES, golden_settings, make_experiment

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_settings_list_passes_through_unchanged():
    from chicdiff_amd import settings as st
    g = golden_settings()
    assert tuple(g.keys()) == st.REFERENCE_KEYS and len(st.REFERENCE_KEYS) == 17     # defaultChicdiffSettings(), chicdiff.R:3-24
    assert not set(st.HIP_KEYS) & set(st.REFERENCE_KEYS)
    s = st.asChicdiffSettings(g)
    assert s["device"] == "png" and st.hipDevice(g) == 0 and st.hipDevice(s) == 0     # the plot device stays the plot device
    assert s["RUexpand"] == 5 and s["score"] == 5.0 and s["norm"] == "combined" and s["theta"] is None
    assert s["theta_grid"] == [0.0, 0.25, 0.5, 0.75, 1.0] and s["saveAuxData"] is False and s["parallel"] is False
    assert list(s["chicagoData"]) == ["CD4", "Mono"] and st.sample_names(s["chicagoData"]) == ["CD4.NCD4_22", "CD4.NCD4_23", "Mono.Mon_2", "Mono.Mon_3"]
    assert st.conditions_per_sample(s["chicagoData"]) == ["CD4", "CD4", "Mono", "Mono"]
    assert st.hipDevice(dict(g, hipDevice=[3])) == 3
    d = st.defaultChicdiffSettings()
    assert tuple(d.keys()) == st.REFERENCE_KEYS and d["device"] == "png" and d["theta_grid"] == [0, 0.25, 0.5, 0.75, 1.0]
    with pytest.raises(ValueError):
        st.asChicdiffSettings(dict(g, devcie="x"))
    with pytest.raises(ValueError):
        st.asChicdiffSettings(dict(g, norm=["loess"]))


def _settings_reads(text):
    return re.findall(r'chicdiff\.settings\[\["([A-Za-z_.]+)"\]\]', text)


def test_r_host_never_reinterprets_a_reference_settings_key():
    """Every chicdiff.settings[["..."]] the R host reads is one of the reference's 17 keys or a declared new one; no
    reference key is coerced to an integer / handed to the GPU context (round 2 read the GPU index from `device`, which
    defaultChicdiffSettings() sets to "png", chicdiff.R:20)."""
    from chicdiff_amd import settings as st
    ref = set(golden_settings().keys())
    rdir = os.path.join(ROOT, "r", "R")
    seen = set()
    for name in sorted(os.listdir(rdir)):
        text = open(os.path.join(rdir, name)).read()
        code = "\n".join(line.split("##")[0] for line in text.splitlines())        # comments may name anything
        for key in _settings_reads(code):
            seen.add(key)
            assert key in ref or key in st.HIP_KEYS, f"{name}: chicdiff.settings[[\"{key}\"]] is neither a reference key nor a declared new key"
        for line in code.splitlines():
            for key in _settings_reads(line):
                if key in ref and key != "RUexpand":
                    assert "as.integer(" not in line and ".hipContext(" not in line and ".hipDeviceIndex" not in line, f"{name}: reference key `{key}` reinterpreted: {line.strip()}"
        assert not re.search(r'\[\["device"\]\][^\n]*(hipContext|as\.integer)', code), name
    # `backend` is read by the lines r/patches/chicdiff_hip.patch adds to the reference's own functions, `hipDevice` by the R host
    patch = open(os.path.join(ROOT, "r", "patches", "chicdiff_hip.patch")).read()
    for line in patch.splitlines():
        if line.startswith("+") and not line.startswith("+++"):
            for key in _settings_reads(line.split("##")[0]):
                seen.add(key)
                assert key in st.HIP_KEYS, f"the patch reads chicdiff.settings[[\"{key}\"]]: only new keys may select the device path"
    assert "hipDevice" in seen and "backend" in seen                                # the GPU index has a key of its own
    # the same for the Python mirrors: no module reads settings["device"]
    for name in os.listdir(os.path.join(ROOT, "chicdiff_amd")):
        if name.endswith(".py") and name != "settings.py":
            assert not re.search(r'\[\s*["\']device["\']\s*\]', open(os.path.join(ROOT, "chicdiff_amd", name)).read()), name
    # and for the documentation a maintainer follows
    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "hipDevice" in integ and not re.search(r'settings\[\["device"\]\]\s*(<-|=)\s*[0-9]', integ)


def test_peak_filter_background_tables_and_distance_function(tmp_path):
    from chicdiff_amd import pipeline, settings as st
    settings, truth = make_experiment(tmp_path, npeaks=1200, with_chinput=False)
    s = st.asChicdiffSettings(settings)
    x = pipeline.readAndFilterPeakMatrix(s["peakfiles"], s["targetColumns"], s["chicagoData"], list(s["chicagoData"]), s["score"],
                                         s["outprefix"])
    assert np.array_equal(x["baitID"].to_numpy(), truth["peak_bait"]) and np.array_equal(x["oeID"].to_numpy(), truth["peak_oe"])
    assert os.path.exists(s["outprefix"] + "_filteredBaits.txt") and 0 < len(x) < 1200
    # the file's layout, pinned: chicdiff.R:272-274 is fwrite(list(filtered_baits), file) on an UNNAMED list — bare IDs,
    # one per line, no header line (data.table >= 1.9.8's fwrite writes column names only when names(x) is a character vector; read
    # from fwriteR.c from memory: no R here, so this pins the mirror's behaviour against silent flips, not data.table's)
    lines = open(s["outprefix"] + "_filteredBaits.txt").read().split("\n")
    assert lines[-1] == "" and all(re.fullmatch(r"[0-9]+", ln) for ln in lines[:-1]), lines[:3]
    import pandas as pd
    every = pd.unique(pd.concat([pd.read_csv(f, sep="\t") for f in (s["peakfiles"] if isinstance(s["peakfiles"], (list, tuple)) else [s["peakfiles"]])])["baitID"])
    assert sorted(int(v) for v in lines[:-1]) == sorted(int(b) for b in every if b not in set(x["baitID"]))
    bg = pipeline.background_tables(truth["xs"], truth["id_min"], truth["nid"])
    for j, t in enumerate(truth["tables"]):
        xs = truth["xs"][j]
        seen_b = np.unique(xs["baitID"]) - truth["id_min"]
        seen_o = np.unique(xs["otherEndID"]) - truth["id_min"]
        assert np.array_equal(bg["sj"][j, seen_b], t["sj"][seen_b], equal_nan=True) and np.isnan(bg["sj"][j, seen_b]).any()
        assert np.array_equal(bg["si"][j, seen_o], t["si"][seen_o])
        unseen = np.setdiff1d(np.arange(truth["nid"]), seen_o)
        assert np.isnan(bg["si"][j, unseen]).all() and (bg["tlb"][j, unseen] == -1).all()
        codes = np.array([bg["levL"].index(v) if v is not None else -1 for v in t["tlb"]])   # codes follow the sorted level names
        assert np.array_equal(bg["tlb"][j, seen_o], codes[seen_o])
        bcodes = np.array([bg["levB"].index(t["bait_tblb"][b + truth["id_min"]]) for b in seen_b])
        assert np.array_equal(bg["tblb"][j, seen_b], bcodes)
        Tt = np.array([[t["T"][t["levB"].index(b), t["levL"].index(l)] for l in bg["levL"]] for b in bg["levB"]])
        ok = ~np.isnan(bg["T"][j])                                                             # a (tblb, tlb) pair no row shows stays NA
        assert ok.mean() > 0.9 and np.array_equal(bg["T"][j][ok], Tt[ok])
    # .chicEstimateDistFun (chicdiff.R:538-573): the fit reproduces an exact cubic, and head / tail continue it with matching value and slope
    import pandas as pd
    mid = 10000.0 + 20000.0 * np.arange(60)
    co = np.array([12.0, -1.3, 0.04, -0.002])
    lm = np.log(mid)
    ref = np.exp(co[0] + co[1] * lm + co[2] * lm ** 2 + co[3] * lm ** 3)
    p = pipeline.chicEstimateDistFun(pd.DataFrame({"distbin": [f"b{k}" for k in range(60)] * 2, "refBinMean": np.tile(ref, 2)}))
    assert np.allclose(p[:4], co, rtol=1e-6) and np.isclose(p[8], lm[0]) and np.isclose(p[9], lm[-1])
    cubic = lambda x: p[0] + p[1] * x + p[2] * x * x + p[3] * x ** 3
    assert np.isclose(p[4] + p[5] * p[8], cubic(p[8])) and np.isclose(p[6] + p[7] * p[9], cubic(p[9]))
    assert np.isclose(p[5], p[1] + 2 * p[2] * p[8] + 3 * p[3] * p[8] ** 2)


def test_dist_lookup_follows_the_reference():
    import pandas as pd
    from chicdiff_amd import pipeline, post
    cov = np.array([np.e ** 2, np.e ** 3, np.e ** 5, np.e ** 6, np.e ** 8, np.e ** 9, 5.0])
    df = pd.DataFrame({"covariate": cov, "group": [1, 1, 2, 2, 3, 3, np.nan]})
    w = np.array([[2.0, 4.0], [1.0, 1.0], [0.5, 0.1]])
    look = pipeline.dist_lookup(df, w)
    assert np.allclose(look["minLogDist"], [0, 5, 8]) and np.allclose(look["maxLogDist"], [3, 6, np.inf])   # [1] <- 0, [n] <- Inf
    assert np.allclose(look["avWeights"], [3.0, 1.0, 0.3])
    assert np.allclose(post.ihw_breaks(look["minLogDist"], look["maxLogDist"]), [0, 4, 7, np.inf])           # chicdiff.R:2039
    with pytest.raises(ValueError):
        pipeline.dist_lookup(pd.DataFrame({"covariate": cov[:2], "group": [1, 3]}), w)                        # "Assumption violated"
