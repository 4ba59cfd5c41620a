import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))

GOLDEN = REPO / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Make sure the native pieces exist (same steps as __graft_entry__.build(): hipcc cross-compiles
    without a GPU, gcc builds the oracle).  Nothing is rebuilt when the artefacts are up to date."""
    from pycamset_amd import build as hip_build

    hip_build.build()
    from oracle import ba_oracle

    ba_oracle.build()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
