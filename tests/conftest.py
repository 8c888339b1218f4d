"""pytest configuration: registers the `gpu` marker and puts the repo root / package dir on sys.path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "self-corr-pose_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """gpu-marked tests SKIP (not fail) where there is no GPU, so `pytest tests/` is meaningful on a CPU box too"""
    import pytest
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a GPU (marked gpu)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
