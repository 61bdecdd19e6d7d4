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


@pytest.fixture(scope="session")
def kats():
    import json

    with open(os.path.join(ROOT, "tests", "golden", "seal_kats.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def moai():
    import __graft_entry__ as g

    return g.load_package()
