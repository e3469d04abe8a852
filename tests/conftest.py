import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_sessionstart(session):
    """The C-ABI library and the CLI are build products (git-ignored).  In a tree where
    `__graft_entry__.build()` has not run yet, build them the same way it does (hipcc
    cross-compiles gfx950 without a GPU) -- the product itself, not a stand-in."""
    lib = os.path.join(REPO, "sipnet_amd", "libsipnet_amd.so")
    cli = os.path.join(REPO, "sipnet_amd", "bin", "sipnet")
    if not (os.path.exists(lib) and os.path.exists(cli)):
        subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "sipnet_amd", "csrc")])


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (test infrastructure), built on demand."""
    from tests import helpers
    return helpers.load_oracle()
