import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# The binding (integration/envelope_vamd.c) lets a stream written in small pieces keep libvorbis' own detector (a GPU round
# trip per sixteen steps costs more than the steps).  The suite's in-process hybrid encodes mostly in 1024-frame writes and
# is there to test the GPU's: force it.  The binding's own choice and the host's are tested through the C application
# (tests/test_dropin_library.py), which runs in processes of its own.
os.environ.setdefault("VAMD_DETECTOR", "gpu")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests never run silently on a box without a GPU
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
