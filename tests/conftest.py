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


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """Make sure the engine and the oracle are built (hipcc cross-compiles without a GPU)."""
    from variantstore_amd import build as vb
    vb.build_all(force=False, verbose=False)   # rebuilds when any source under csrc/ is newer than the library
    from oracle import oracle as orc
    orc._load()
    return True


@pytest.fixture(scope="session")
def survey_vectors():
    with open(os.path.join(GOLDEN, "survey_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
