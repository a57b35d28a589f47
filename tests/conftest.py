import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a box without a HIP device: the @pytest.mark.gpu tests are skipped instead of failing (the product has no
    CPU path; `-m gpu` on the GPU box runs them)."""
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:
        have = False
    if have:
        return
    import pytest
    skip = pytest.mark.skip(reason="needs a HIP device (MI355X)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
