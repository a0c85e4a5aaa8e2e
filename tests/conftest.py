import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _have_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


HAVE_GPU = None


def have_gpu():
    global HAVE_GPU
    if HAVE_GPU is None:
        HAVE_GPU = _have_gpu()
    return HAVE_GPU


def pytest_collection_modifyitems(config, items):
    # -m gpu on a box without a GPU must fail loudly, not skip: the product has no fallback.
    pass


@pytest.fixture(scope="session")
def oracle():
    from tests.oracle_bind import Oracle

    return Oracle()


@pytest.fixture(scope="session")
def mf():
    lib = os.path.join(ROOT, "matrixfactorizationsgd.java_amd", "lib", "libmfsgd.so")
    if not os.path.exists(lib):
        subprocess.run(["make", "-C", os.path.join(ROOT, "matrixfactorizationsgd.java_amd", "csrc")], check=True,
                       stdout=subprocess.DEVNULL)
    import mfsgd_amd

    mfsgd_amd.load_library()
    return mfsgd_amd
