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
    """Paths of the HIP C-ABI libraries; built here when missing (hipcc cross-compiles)."""
    from sppark_amd import build as B
    from sppark_amd import ffi
    missing = [n for n in ("bls12_381", "bn254", "gl64", "bb31") if not os.path.exists(ffi.lib_path(n))]
    if missing:
        B.build(only=missing, verbose=False)
    return {n: ffi.lib_path(n) for n in ("bls12_381", "bn254", "gl64", "bb31")}


def have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
