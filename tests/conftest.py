import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; oracle/ddgi_oracle.c)."""
    from oracle import oracle_py

    oracle_py.build()
    oracle_py.set_arith(True)
    return oracle_py


@pytest.fixture(scope="session")
def ddgi():
    """The product package (ctypes binding of libddgi_probe.so)."""
    import ddgi_amd

    ddgi_amd.load_library()
    return ddgi_amd


@pytest.fixture(autouse=True)
def _pinned_arith_by_default():
    # every test starts in PINNED arithmetic; tests that switch to LITERAL restore it here
    yield
    try:
        from oracle import oracle_py

        if oracle_py._lib is not None:
            oracle_py.set_arith(True)
            oracle_py.set_ray_tile(0, 0)
    except Exception:
        pass
