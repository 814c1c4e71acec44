import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Build the HIP library and the oracle in-tree if a fresh checkout has neither (hipcc
    cross-compiles gfx950 without a GPU; about half a minute). The GPU box gets prebuilt files."""
    import subprocess

    lib = os.path.join(ROOT, "fft_amd", "lib", "libthunderfft_hip.so")
    ora = os.path.join(ROOT, "oracle", "liboracle_dft.so")
    ex = os.path.join(ROOT, "examples", "libspectral_mask.so")
    ex2 = os.path.join(ROOT, "examples", "libreference_shape.so")
    if not (os.path.exists(lib) and os.path.exists(ora) and os.path.exists(ex) and os.path.exists(ex2)):
        subprocess.check_call(["make", "-C", ROOT, "-j8", "lib", "oracle", "examples"], stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle

    oracle.build()
    return oracle
