"""CPU-only checks of the drop-in boundary: the C-ABI library builds, loads, exports every
symbol include/chicdiff_hip.h declares, and refuses (loudly) to run without a GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from chicdiff_amd import hip
    return hip.load_library()


def test_exports_match_header(lib):
    from chicdiff_amd import hip
    hdr = open(os.path.join(ROOT, "include", "chicdiff_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(chicdiff_hip_[a-z0-9_]+)\s*\(", hdr)))
    assert declared == sorted(hip.EXPORTS)
    for sym in declared:
        assert hasattr(lib, sym), sym


def test_struct_layouts_match_header(lib):
    from chicdiff_amd import hip
    o = hip.default_opts()
    assert (o.minDisp, o.dispTol, o.kappa0, o.maxit, o.betaMaxit, o.betaTol, o.minmu, o.outlierSD) == \
        (1e-8, 1e-6, 1.0, 100, 100, 1e-8, 0.5, 2.0)
    assert o.dispPriorVar != o.dispPriorVar and o.trendCoef[0] != o.trendCoef[0]  # NaN = estimate
    assert C.sizeof(hip.Opts) == 88 and o.fitType == 0 and C.sizeof(hip.Out) == 21 * 8 and C.sizeof(hip.Scalars) == 56
    # the compiler's view of the header (gcc, plain C): sizes and a few offsets against the ctypes mirrors
    import subprocess
    import tempfile
    src = r"""
#include <stddef.h>
#include <stdio.h>
#include "chicdiff_hip.h"
int main(void) {
    printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(chicdiff_nbglm_opts), sizeof(chicdiff_nbglm_out), sizeof(chicdiff_nbglm_scalars),
           sizeof(chicdiff_results_info), offsetof(chicdiff_results_info, index), offsetof(chicdiff_results_info, theta),
           offsetof(chicdiff_nbglm_opts, trendCoef));
    return 0;
}
"""
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "t.c"), "w").write(src)
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(td, "t.c"), "-o", os.path.join(td, "t")], check=True)
        got = [int(x) for x in subprocess.run([os.path.join(td, "t")], capture_output=True, text=True, check=True).stdout.split()]
    assert got == [C.sizeof(hip.Opts), C.sizeof(hip.Out), C.sizeof(hip.Scalars), C.sizeof(hip.ResultsInfo), hip.ResultsInfo.index.offset,
                   hip.ResultsInfo.theta.offset, hip.Opts.trendCoef.offset]


def test_product_path_fails_loudly_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from chicdiff_amd import hip
    with pytest.raises(hip.ChicdiffHipError):
        hip.HipContext(0)
    h = C.c_void_p()
    rc = lib.chicdiff_hip_create(C.byref(h), 0)
    assert rc != 0 and not h.value
    assert b"no CPU fallback" in lib.chicdiff_hip_last_error(None)


def test_product_package_never_imports_oracle():
    pat = re.compile(r"^\s*(from\s+oracle|import\s+oracle|#include\s+[\"<].*oracle)", re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "chicdiff_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                assert not pat.search(open(os.path.join(dirpath, f)).read()), f
