import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

# The HIP runtime reports fatal conditions (a GPU memory fault, a queue error) through its log and then abort()s; with the
# default log level 0 that is a silent "Fatal Python error: Aborted" (seen in 1 of 14 full suite runs of round 5, cause open:
# DESIGN.md section 9).  Level 1 = errors only: silent in normal operation (checked), and a rare abort names its reason.
os.environ.setdefault("AMD_LOG_LEVEL", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
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
