"""orc_redistribute_threads (oracle/lentil_oracle.cpp): the oracle's visits over several threads and ONE frame -- what the 4K
parity tests and bench.py's cpu_baseline run -- against the same visits on one thread (orc_redistribute, the restatement of
src/lentil_filter.cpp:105-448 proper): counters and draw log identical, pixels no draw reaches bit for bit, closest-filtered
AOVs (src/lentil.h:832-837) bit for bit, gaussian sums to fp32 summation order."""
import numpy as np
import pytest

import common
import oracle_lib


def _both(orc, p, table, visits, n_aovs, kinds, n_threads):
    import ctypes as C
    lens = orc.orc_lens_create(C.byref(table)) if table is not None else None
    one = oracle_lib.Frame(orc, p, n_aovs=n_aovs, kinds=kinds, keep_log=True)
    one.run(lens, None, visits)
    many = oracle_lib.Frame(orc, p, n_aovs=n_aovs, kinds=kinds, keep_log=True)
    assert many.run_threads(lens, None, visits, n_threads, visits.pixels_per_row * visits.visits_per_pixel)
    if lens:
        orc.orc_lens_destroy(lens)
    return one, many


@pytest.mark.parametrize("camera,n_threads", [("po", 5), ("po", 64), ("thinlens", 3)])
def test_threads_equal_one_thread(orc, camera, n_threads):
    W, H, M = 96, 64, 9
    if camera == "po":
        p, model, table, keep = common.po_setup(W, H, samples_override=48)
    else:
        p, table = common.tl_setup(W, H, samples_override=48), None
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.02, n_extra=2)
    kinds = [0, 1, 0]
    one, many = _both(orc, p, table, visits, 3, kinds, n_threads)
    try:
        c1, c2 = one.counters(), many.counters()
        assert (c1.visits, c1.redistributed_visits, c1.attempted_draws, c1.accepted_draws) == (
            c2.visits, c2.redistributed_visits, c2.attempted_draws, c2.accepted_draws)
        assert c1.accepted_draws > 1000
        assert np.array_equal(one.log(), many.log())                   # same records in the same (visit) order
        touched = np.zeros(p.xres * p.yres, bool)
        touched[one.log()[:, 2]] = True
        assert touched.any() and not touched.all()
        for a in (0, 2):
            b1, b2 = one.buffer(a), many.buffer(a)
            assert np.array_equal(b1[~touched], b2[~touched])
            e1, e2 = one.buffer64(a), many.buffer64(a)
            m = e1 != 0
            assert np.array_equal(m, e2 != 0)
            assert float(np.max(np.abs(e1[m] - e2[m]) / np.abs(e1[m]))) < 1e-12
            assert float(np.max(np.abs(b2[m].astype(np.float64) - e1[m]) / np.abs(e1[m]))) < 1e-5
        assert np.array_equal(one.weight()[~touched], many.weight()[~touched])
        assert np.array_equal(one.buffer(1), many.buffer(1))            # the closest-filtered AOV: a copy of one candidate
        assert np.array_equal(one.zbuffer(), many.zbuffer()) and np.array_equal(one.zvisit(), many.zvisit())
        assert np.array_equal(one.resolve(1), many.resolve(1))
    finally:
        one.close()
        many.close()


def test_what_the_threaded_form_does_not_take(orc):
    """thin lens with abb_chromatic > 0 draws its colour channels from ONE xor128 stream in visit order (src/lentil_filter.cpp:397):
    not to be split over threads -- refused, and common.ThreadedOracle falls back to private frames."""
    W, H, M = 32, 24, 9
    p = common.tl_setup(W, H, samples_override=16, abb_chromatic=0.5)
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.02)
    fr = oracle_lib.Frame(orc, p, keep_log=True)
    try:
        assert fr.run_threads(None, None, visits, 4, W * M) is False
        assert fr.counters().visits == 0
    finally:
        fr.close()
