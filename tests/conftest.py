import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "gpu_soak: seeded soaks / randomised sweeps / child-process runs on a GPU -- not part of -m gpu "
                                       "(whose run has a time limit): tools/soak.sh runs them")


@pytest.fixture(scope="session", autouse=True)
def _built():
    import __graft_entry__ as g
    g.build()
    yield
    # A streamed pass whose resident waves give up waiting is run again and its result is right -- so nothing but this would
    # ever fail on a stall: at the end of the session every time-out the library counted must be one a test asked for
    # (LENTIL_INJECT_STALL).  (Contexts are closed by now; the count is the process's.)
    try:
        from pota_amd import capi
        streamed, stuck, injected, redone = capi.process_stats()
    except Exception:          # (no library: the tests that need it have said so)
        return
    import stalls
    assert stuck == injected + stalls.tolerated, ("%d of this session's %d streamed passes hit the stuck time-out without a test asking for it "
                               "(%d asked for; %d passes wiped and run again): a stall.  What the waves that gave up saw:\n%s"
                               % (stuck - injected - stalls.tolerated, streamed, injected + stalls.tolerated, redone, capi.process_stall_notes()))


@pytest.fixture(autouse=True)
def _no_unasked_stall(request):
    """A streamed pass whose waves give up waiting is redone and right, so only this makes the test that met it fail: every
    stuck time-out the library counts during a test must be one the test asked for (LENTIL_INJECT_STALL) or said it tolerates
    (tests/stalls.py)."""
    if request.node.get_closest_marker("gpu") is None and request.node.get_closest_marker("gpu_soak") is None:
        yield
        return
    import stalls
    from pota_amd import capi
    try:
        before = capi.process_stats()
    except Exception:          # noqa: BLE001
        yield
        return
    tol0 = stalls.tolerated
    yield
    after = capi.process_stats()
    unasked = (after[1] - before[1]) - (after[2] - before[2]) - (stalls.tolerated - tol0)
    if unasked > 0:
        stalls.tolerated += unasked          # (reported here: the session's closing assertion need not repeat it)
        pytest.fail("%d streamed pass(es) of this test hit the stuck time-out unasked.  What the waves that gave up saw:\n%s"
                    % (unasked, capi.process_stall_notes()))


@pytest.fixture(scope="session")
def orc():
    import oracle_lib
    return oracle_lib.load()


# (function scope: the contexts a test makes go with the test.  Kept alive for the whole session -- as they were until round 6 --
# they were dozens by the end, each with its six streams, and the runtime spreads a process's streams over its hardware queues
# as they are created: late tests' passes then shared queues, their resident kernels sat behind the ones they waited for, and
# a streamed pass now and then hit the stuck time-out and was redone -- correct, 250 ms late, and invisible until the library
# started counting: three such passes in one run of the suite, all in tests near its end.)
@pytest.fixture
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
