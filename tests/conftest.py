import os
import sys

import pytest

# One HIP runtime per process: PyTorch wheels bundle their own libamdhip64, and whichever copy initialises
# the device first wins -- a second one then reports "No HIP GPUs are available".  Tests that pass torch
# tensors / streams to the C ABI therefore need torch imported BEFORE libmoai_hip.so is loaded.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch is optional for everything but the bench and the graph test
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout holds no built artefacts (they are git-ignored): build them once where a compiler is
    at hand, so that `pytest tests` works without a separate build step.  Where hipcc is absent the tests that
    need the library fail loudly, as they should."""
    import shutil

    pkg = os.path.join(ROOT, "moai-fhe-transformerinference-public_amd")
    needed = [os.path.join(pkg, "libmoai_hip.so"), os.path.join(ROOT, "oracle", "libmoai_oracle.so"),
              os.path.join(ROOT, "tests", "cpp", "test_seal_shim")]
    if all(os.path.exists(p) for p in needed):
        return
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        return
    import __graft_entry__ as g

    g.build()


@pytest.fixture(scope="session")
def kats():
    import json

    with open(os.path.join(ROOT, "tests", "golden", "seal_kats.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def moai():
    import __graft_entry__ as g

    return g.load_package()
