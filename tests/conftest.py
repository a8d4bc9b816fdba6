import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    import __graft_entry__ as g
    g.build()


@pytest.fixture(scope="session")
def orc():
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def gpu_ctx_factory():
    from pota_amd import capi

    made = []

    def make():
        c = capi.Context(0)
        made.append(c)
        return c

    yield make
    for c in made:
        c.close()
