"""a13 on the GPU (Camera::lens_evaluate, Camera::lens_pt_sample_aperture, src/lentil.h:1257-1291) and the focus search
built on them (Camera::camera_get_y0_intersection_distance :1361-1386, Camera::logarithmic_focus_search :1445-1460):
one lane per candidate sensor shift, against the oracle's sequential loop -- fp64, bit for bit."""
import ctypes as C

import numpy as np
import pytest

import common
import oracle_lib

pytestmark = pytest.mark.gpu
LENSES = ["double_gauss_50mm", "petzval_58mm", "anamorphic_petzval_58mm"]


def _oracle_chain(orc, lens, table, shift, lam):
    """what camera_get_y0_intersection_distance does, step by step, in the oracle"""
    sensor = (C.c_double * 5)(0, 0, 0, 0, lam)
    ap = (C.c_double * 5)(0, float(table.lens_aperture_housing_radius) * 0.25, 0, 0, 0)
    out = (C.c_double * 5)()
    orc.orc_pt_sample_aperture(lens, sensor, ap, shift)
    sensor[0] += sensor[2] * shift
    sensor[1] += sensor[3] * shift
    T = orc.orc_lens_evaluate(lens, sensor, out)
    return list(sensor), [out[0], out[1], out[2], out[3], T]


@pytest.mark.parametrize("lens_name", LENSES)
def test_pt_sample_aperture_and_lens_evaluate_bit_exact(orc, gpu_ctx_factory, lens_name):
    p, model, table, keep = common.po_setup(64, 48, lens=lens_name)
    ctx = gpu_ctx_factory()
    ctx.set_params(p); ctx.set_lens(table)
    rng = np.random.default_rng(13)
    shifts = np.concatenate([rng.uniform(-45.0, 45.0, 1500), rng.uniform(-2.0, 2.0, 500), [0.0, -45.0, 45.0]])
    lens = orc.orc_lens_create(C.byref(table))
    for lam in (0.55, 0.45, float(np.float32(0.55))):
        dist, sensor, out = ctx.test_y0_intersection(shifts, lam)
        es, eo, ed = np.empty_like(sensor), np.empty_like(out), np.empty_like(dist)
        for i, sh in enumerate(shifts):
            es[i], eo[i] = _oracle_chain(orc, lens, table, float(sh), lam)
            ed[i] = orc.orc_camera_get_y0_intersection_distance(lens, float(sh), lam)
        assert np.array_equal(sensor, es, equal_nan=True)          # lens_pt_sample_aperture: the solved direction
        assert np.array_equal(out, eo, equal_nan=True)             # lens_evaluate: out x, y, dx, dy, max(0, transmittance)
        assert np.array_equal(dist, ed, equal_nan=True)
        assert np.isfinite(ed).mean() > 0.9
    orc.orc_lens_destroy(lens)


@pytest.mark.parametrize("lens_name", LENSES)
def test_focus_search_matches_the_sequential_loop(orc, gpu_ctx_factory, lens_name):
    """The winner of 20 001 candidates: the sensor shift the reference's loop ends with, for focus distances from
    30 cm to "infinity" (the reference also searches 999999999.0, src/lentil.h:1643), and the shift the camera setup
    puts into the parameters (liblentil_host's camera_model_specific_setup, itself bit-identical to the oracle)."""
    p, model, table, keep = common.po_setup(64, 48, lens=lens_name, focus_dist=150.0)
    ctx = gpu_ctx_factory()
    ctx.set_params(p); ctx.set_lens(table)
    lens = orc.orc_lens_create(C.byref(table))
    seen = set()
    for focal_mm in (300.0, 1500.0, 1234.5, 8000.0, 50000.0, 999999999.0):
        for lam in (0.55, 0.62):
            want = orc.orc_logarithmic_focus_search(lens, focal_mm, lam)
            got = ctx.focus_search(focal_mm, lam)
            assert got == want, (lens_name, focal_mm, lam, got, want)
            seen.add(want)
    assert len(seen) >= 4                      # the cases do pick different candidates
    orc.orc_lens_destroy(lens)
    # the shift of this very camera: focus_distance is in cm on the parameter, mm in the search (src/lentil.h:1573)
    assert ctx.focus_search(float(p.focus_distance), 550.0 * 0.001) == float(p.sensor_shift)
