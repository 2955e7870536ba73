import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name + ".json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session")
def gpu_ctx():
    """A common_amd.Context on cuda:0; only -m gpu tests ask for it."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("test marked gpu but no GPU is visible")
    import common_amd
    return common_amd.Context(device=0)
