import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def engine():
    """One context on cuda:0 shared by the GPU tests (fails loudly if the library or GPU is missing)."""
    import torch  # noqa: F401  (first: shares libamdhip64 with the extension)
    import cpprob_amd
    eng = cpprob_amd.Engine(0)
    yield eng
    eng.close()
