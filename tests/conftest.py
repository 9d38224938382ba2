import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def libs():
    """Paths of the HIP C-ABI libraries.  Always goes through sppark_amd.build: a no-op when the
    source-hash stamp of every library matches csrc/, a rebuild otherwise (hipcc cross-compiles
    here; on the GPU box the libraries arrive prebuilt with matching stamps), so the tests never
    run against binaries that were not built from the tree they sit in."""
    from sppark_amd import build as B
    from sppark_amd import ffi
    B.build(verbose=False)
    return {n: ffi.lib_path(n) for n in B.PRODUCT}


def have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
