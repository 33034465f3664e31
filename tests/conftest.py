import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    """The HIP C-ABI library; built in-tree by __graft_entry__.build()."""
    import torch  # noqa: F401  (first, so that both share one HIP runtime)
    from xrsfm_amd import _build, capi
    _build.build_lib()
    return capi.load()
