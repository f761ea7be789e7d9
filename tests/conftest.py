import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # test infrastructure only: run the suite against a variant build of the library (A/B kernels under development).
    # The package itself reads no environment; the choice is handed over explicitly.
    variant = os.environ.get("SGG_TEST_LIB")
    if variant:
        import sggan_amd
        sggan_amd._abi.use_library(variant if os.path.isabs(variant) else os.path.join(ROOT, "sg-gan-tf2_amd", variant))


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
