import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "chr19_results.npz"), allow_pickle=True)


def pytest_sessionfinish(session, exitstatus):
    """GPU runs: keep the numbers behind every parity assertion (max relative error, rows off) where gpurun brings
    them back (gpurun_out/), to be copied into profiles/ with the round's other evidence."""
    import json
    mod = sys.modules.get("test_gpu_parity") or sys.modules.get("tests.test_gpu_parity")
    log = getattr(mod, "PARITY_LOG", None) if mod else None
    if log:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_gpu.json"), "w") as f:
            json.dump(dict(exitstatus=int(exitstatus), comparisons=log), f, indent=1)
