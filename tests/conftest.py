import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def nso():
    import nso as _nso
    return _nso


@pytest.fixture(scope="session")
def refk(nso):
    r = nso.ref()
    if r is None:
        pytest.skip("oracle/_ref/libkernel_ref.so not built (reference tree absent)")
    return r


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as ge
    p = ge.load_package()
    import os as _os
    if not _os.path.exists(p.LIB_PATH):
        p.build()
    return p


@pytest.fixture(scope="session")
def L(pkg):
    return pkg.lib()
