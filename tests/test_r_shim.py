"""The R host (r/) cannot run here — no R in the image.  These CPU checks keep it honest as far as a
C compiler and text can: the shim compiles against declaration-only stand-ins for R's headers (tests/r_stub/ — a
syntax / prototype check that PINS NOTHING about R's behaviour), every .Call in the R sources names a registered
routine with the registered number of arguments, every library entry point the shim uses is declared in
include/chicdiff_hip.h, and the R file defines the reference's DESeq2Wrap signature."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "r", "src", "chicdiff_hip_shim.c")
RSRC = [os.path.join(ROOT, "r", "R", "DESeq2Wrap_hip.R"), os.path.join(ROOT, "tools", "make_golden.R"),
        os.path.join(ROOT, "r", "R", "getFullRegionData_hip.R"), os.path.join(ROOT, "r", "R", "post_hip.R")]


def test_shim_compiles_against_declaration_only_r_headers():
    r = subprocess.run(["gcc", "-std=gnu99", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-Wno-cast-function-type",
                        "-I", os.path.join(ROOT, "tests", "r_stub"), "-I", os.path.join(ROOT, "include"), SHIM],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def _registered():
    src = open(SHIM).read()
    reg = {m.group(1): int(m.group(2)) for m in re.finditer(r'\{"(chicdiff_hip_\w+)", \(DL_FUNC\)&\1, (\d+)\}', src)}
    # the definition of each registered routine takes that many SEXP arguments
    for name, nargs in reg.items():
        m = re.search(r"^SEXP " + name + r"\(([^)]*)\)", src, re.M | re.S)
        assert m, name
        assert m.group(1).count("SEXP") == nargs, (name, m.group(1))
    return reg


def _calls(text):
    """(.Call name, number of arguments after the name, PACKAGE excluded) for every .Call in an R source; .hipCall(name, ...)
    is the wrapper that appends PACKAGE itself."""
    out = []
    for m in re.finditer(r'\.(?:hip)?Call\("(\w+)"', text):
        i, depth, nargs, in_str = m.end(), 1, 0, None
        while depth:
            ch = text[i]
            if in_str:
                if ch == "\\":
                    i += 1
                elif ch == in_str:
                    in_str = None
            elif ch in "\"'":
                in_str = ch
            elif ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
            elif ch == "," and depth == 1:
                nargs += 1
            i += 1
        if text[m.start():].startswith(".hipCall"):
            out.append((m.group(1), nargs))
            continue
        assert "PACKAGE" in text[m.start():i]
        out.append((m.group(1), nargs - 1))  # the trailing PACKAGE = argument
    return out


def test_every_dot_call_is_registered_with_matching_arity():
    reg = _registered()
    assert len(reg) >= 18
    seen = set()
    for path in RSRC:
        if not os.path.exists(path):
            continue
        for name, nargs in _calls(open(path).read()):
            assert name in reg, (path, name)
            assert reg[name] == nargs, (path, name, nargs, reg[name])
            seen.add(name)
    for must in ("chicdiff_hip_open", "chicdiff_hip_window_sums", "chicdiff_hip_size_factors", "chicdiff_hip_theta_grid",
                 "chicdiff_hip_wald_test", "chicdiff_hip_fit", "chicdiff_hip_release", "chicdiff_hip_chinput_table",
                 "chicdiff_hip_count_join", "chicdiff_hip_fragment_background", "chicdiff_hip_alloc", "chicdiff_hip_region_universe",
                 "chicdiff_hip_ihw_apply", "chicdiff_hip_bait_flags", "chicdiff_hip_count_table", "chicdiff_hip_count_join_inner",
                 "chicdiff_hip_region_avdist", "chicdiff_hip_download", "chicdiff_hip_upload"):
        assert must in seen, must


def test_shim_uses_only_declared_library_entry_points():
    hdr = open(os.path.join(ROOT, "include", "chicdiff_hip.h")).read()
    declared = set(re.findall(r"\b(chicdiff_hip_[a-z0-9_]+)\s*\(", hdr))
    src = open(SHIM).read()
    reg = set(_registered())
    used = set(re.findall(r"\b(chicdiff_hip_[a-z0-9_]+)\s*\(", src)) - reg
    assert used and used <= declared, used - declared
    for must in ("chicdiff_hip_window_sums_dev", "chicdiff_hip_size_factors_dev", "chicdiff_hip_offsets_dev", "chicdiff_hip_theta_grid_dev",
                 "chicdiff_hip_wald_test_dev", "chicdiff_hip_nbglm_fit_dev", "chicdiff_hip_cooks_filter_dev",
                 "chicdiff_hip_independent_filtering_dev", "chicdiff_hip_chinput_read", "chicdiff_hip_chinput_table_dev",
                 "chicdiff_hip_count_join_dev", "chicdiff_hip_fragment_background_dev", "chicdiff_hip_count_table_dev",
                 "chicdiff_hip_count_join_inner_dev", "chicdiff_hip_region_avdist_dev"):
        assert must in used, must


def test_r_wrapper_defines_the_reference_signature_and_messages():
    r = open(RSRC[0]).read()
    assert re.search(r'^DESeq2Wrap <- function\(chicdiff\.settings, RU, FullRegionData, suffix = "", theta = NULL\)', r, re.M)
    for text in ("DESeq2Wrap error: Unknown normalisation method.", "Optimising scaling factors...",
                 "Total deviances by theta (Fullmean --> Standard):", "Theta=", "Processing model output",
                 ": # unweighted interactions with padj<0.05: ", "Standard DESeq2 normalisation",
                 "Chicago full mean-based normalisation", "combined normalisation",
                 'Mixing parameter theta set to 1, equivalent to norm = \\"standard\\". The norm method has been reset accordingly.',
                 'Mixing parameter theta set to 0, equivalent to norm = \\"fullmean\\". The norm method has been reset accordingly.'):
        assert text in r, text
    assert "unseeded" not in r and "session RNG" not in r
    # the stages either side keep the reference's signatures too (chicdiff.R:1460, :1956), so chicdiffPipeline() (:301-347)
    # reaches the device path without a change of its own
    g = open(RSRC[2]).read()
    assert re.search(r'^getFullRegionData <- function\(chicdiff\.settings, RU, RUcontrol, suffix = ""\)', g, re.M)
    p = open(RSRC[3]).read()
    assert re.search(r"^IHWcorrection <- function\(chicdiff\.settings, DESeqOut, FullRegionData, DESeqOutControl, FullControlRegionData,\s*countput, DiagPlot = TRUE, diffbaitPlot = TRUE, suffix = \"\"\)", p, re.M)
    assert re.search(r'^getRegionUniverse <- function\(chicdiff\.settings, suffix = ""\)', p, re.M)
