"""The reference's settings list as the Python host mirrors see it.

``chicdiff.settings`` is the 17-key named list of ``defaultChicdiffSettings()`` (chicdiff.R:3-24), filled in by
``setChicdiffExperiment()`` (:31-48; stays reference R — SURVEY.md §2 row 1).  The mirrors take that list as a dict
— as exported from R (e.g. ``jsonlite::toJSON`` / the committed fixture ``tests/golden/chr19_settings.json``, where
every scalar arrives as a one-element list) or written by hand — and never reinterpret one of its keys: the
reference's ``device`` is the *graphics* device of the diagnostic plots (``"png"``, chicdiff.R:20, :1960, :2058), so
the GPU index lives under a key of its own.

New, optional keys (absent = default), all outside the reference's 17:
  ``backend``        ``"hip"`` selects the device path in the R host (r/R/*.R); the Python mirrors are the device path.
  ``hipDevice``      GPU index of this process (default 0).
  ``trendFallback``  ``"mean"``: refit with DESeq2's fitType = "mean" when no dispersion trend can be fitted at all.
"""
from __future__ import annotations

from collections import OrderedDict

# defaultChicdiffSettings(), chicdiff.R:3-24 — names and order
REFERENCE_KEYS = ("inputfiles", "peakfiles", "chicagoData", "countData", "rmapfile", "targetColumns", "baitmapfile",
                  "RUexpand", "score", "norm", "theta", "theta_grid", "saveAuxData", "parallel", "device", "printMemory",
                  "outprefix")
HIP_KEYS = ("backend", "hipDevice", "trendFallback")
_VECTOR_KEYS = ("theta_grid", "targetColumns", "peakfiles")   # stay lists whatever their length
_LOGICAL_KEYS = ("saveAuxData", "parallel", "printMemory")


def defaultChicdiffSettings() -> dict:
    """chicdiff.R:3-24 (NA -> None)."""
    return dict(inputfiles=None, peakfiles=None, chicagoData=None, countData=None, rmapfile=None, targetColumns=None,
                baitmapfile=None, RUexpand=5, score=5, norm="combined", theta=None, theta_grid=[0, 0.25, 0.5, 0.75, 1.0],
                saveAuxData=False, parallel=False, device="png", printMemory=False, outprefix="")


def _unbox(v):
    """An R scalar exported as a one-element list -> the scalar; NA / null -> None."""
    if isinstance(v, (list, tuple)) and len(v) == 1:
        v = v[0]
    if isinstance(v, float) and v != v:
        return None
    return v


def _file_lists(v):
    """chicagoData / countData: list(<condition> = c(<replicate> = "<path>", ...), ...) (chicdiff.R:91-114, :177-193)
    -> OrderedDict condition -> OrderedDict replicate name -> path.  A bare list of names stands for name == path."""
    v = _unbox(v) if not isinstance(v, dict) else v
    if v is None:
        return None
    if not isinstance(v, dict):
        raise ValueError("chicagoData / countData must map each condition to its replicates' files")
    out = OrderedDict()
    for cond, files in v.items():
        if isinstance(files, dict):
            out[str(cond)] = OrderedDict((str(k), str(p)) for k, p in files.items())
        else:
            files = [files] if isinstance(files, str) else list(files)
            out[str(cond)] = OrderedDict((str(p), str(p)) for p in files)
    return out


def asChicdiffSettings(settings: dict) -> dict:
    """The settings dict in one canonical form; every reference key keeps the reference's meaning and type
    (``device`` stays the plot device).  Unknown keys raise: a typo must not silently become a default."""
    unknown = [k for k in settings if k not in REFERENCE_KEYS and k not in HIP_KEYS]
    if unknown:
        raise ValueError(f"unknown chicdiff.settings keys: {unknown}")
    s = defaultChicdiffSettings()
    for k, v in settings.items():
        if k in ("chicagoData", "countData"):
            s[k] = _file_lists(v)
        elif k in _VECTOR_KEYS:
            v = _unbox(v) if not isinstance(v, (list, tuple)) else v
            s[k] = None if v is None else ([v] if isinstance(v, (str, int, float)) else list(v))
            if k == "peakfiles" and s[k] == [None]:
                s[k] = None
        else:
            s[k] = _unbox(v)
    for k in _LOGICAL_KEYS:
        s[k] = bool(s[k])
    s["RUexpand"] = int(s["RUexpand"])
    s["norm"] = str(s["norm"]).lower()                       # chicdiff.R:149
    if s["norm"] not in ("standard", "fullmean", "combined"):
        raise ValueError("Parameter error: normalisation method should be one of 'standard', 'fullmean', 'combined'")
    if s["chicagoData"] is not None and s["countData"] is not None:
        if list(s["chicagoData"]) != list(s["countData"]):   # chicdiff.R:100-102
            raise ValueError("Conditions for the RDS/RDA files and chinputs must be the same")
        if sum(map(len, s["chicagoData"].values())) != sum(map(len, s["countData"].values())):   # :103-107
            raise ValueError("Must provide the same number of RDS/RDA files as chinputs")
    if s.get("hipDevice") is not None:
        s["hipDevice"] = int(s["hipDevice"])
    return s


def hipDevice(settings: dict) -> int:
    """GPU index of this process: the optional ``hipDevice`` key — never ``device`` (the reference's plot device)."""
    v = _unbox(settings.get("hipDevice"))
    return 0 if v is None else int(v)


def sample_names(file_lists) -> list:
    """names(unlist(chicagoData)) in R's order: "<condition>.<replicate>" (chicdiff.R:586, :917)."""
    return [f"{c}.{r}" for c, reps in file_lists.items() for r in reps]


def conditions_per_sample(file_lists) -> list:
    """rep(names(chicagoData), sapply(chicagoData, length)) (chicdiff.R:921-923)."""
    return [c for c, reps in file_lists.items() for _ in reps]
