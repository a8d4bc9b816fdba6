"""Host camera setup (pota_amd/camera.py) against the oracle: lens_evaluate, lens_pt_sample_aperture
(a13) and the quantities derived from them."""
import ctypes as C

import numpy as np

import common
from oracle_lib import darr
from pota_amd import camera, lens_io


def test_lens_evaluate_matches_oracle_bitwise(orc):
    model = camera.LensModel("double_gauss_50mm")
    table, keep = lens_io.make_lens_table(model.spec)
    lens = orc.orc_lens_create(C.byref(table))
    rng = np.random.default_rng(0)
    pts = np.stack([rng.uniform(-15, 15, 200), rng.uniform(-15, 15, 200), rng.uniform(-0.2, 0.2, 200),
                    rng.uniform(-0.2, 0.2, 200)], 1)
    out, T = model.lens_evaluate(pts[:, 0], pts[:, 1], pts[:, 2], pts[:, 3], 0.55)
    o5 = (C.c_double * 5)()
    for i in range(pts.shape[0]):
        t = orc.orc_lens_evaluate(lens, darr(*pts[i], 0.55), o5)
        assert [out[k][i] for k in range(4)] == [o5[k] for k in range(4)]
        assert T[i] == t
    orc.orc_lens_destroy(lens)


def test_pt_sample_aperture_matches_oracle_bitwise(orc):
    model = camera.LensModel("double_gauss_50mm")
    table, keep = lens_io.make_lens_table(model.spec)
    lens = orc.orc_lens_create(C.byref(table))
    rng = np.random.default_rng(1)
    for _ in range(50):
        ax, ay = rng.uniform(-4, 4, 2)
        x, y = rng.uniform(-10, 10, 2)
        dist = rng.uniform(0, 5)
        dx, dy, pdx, pdy = model.pt_sample_aperture(x, y, 0.0, 0.0, 0.55, ax, ay, dist)
        inn = darr(x, y, 0.0, 0.0, 0.55)
        out = darr(ax, ay, 0.0, 0.0, 0.0)
        orc.orc_pt_sample_aperture(lens, inn, out, dist)
        assert (float(dx), float(dy)) == (inn[2], inn[3])
        assert (float(pdx), float(pdy)) == (out[2], out[3])
    orc.orc_lens_destroy(lens)


def test_focus_search_focuses(orc):
    """The sensor shift found by the logarithmic search (paraxial: it traces the ray through the aperture
    point (0, housing_radius/4), src/lentil.h:1361-1386) focuses the requested distance: an on-axis
    point on the focus plane is traced back to one sensor point for small aperture offsets, points
    nearer / farther are not."""
    p, model, table, keep = common.po_setup(64, 48, focus_dist=150.0)
    assert 0.5 < p.sensor_shift < 6.0
    lens = orc.orc_lens_create(C.byref(table))
    s5, o5 = (C.c_double * 5)(), (C.c_double * 5)()

    def spread(z):
        ys = []
        for ap in [(0.0, 0.0), (0.0, 1.2), (0.0, -1.2)]:
            T = orc.orc_lt_sample_aperture(lens, darr(0.0, 0.0, z), darr(*ap), s5, o5, 0.55, None)
            assert T > 0
            ys.append(s5[1] - s5[3] * p.sensor_shift)        # src/lentil.h:654-655
        return max(ys) - min(ys)

    in_focus = spread(1500.0)
    assert in_focus < 2e-3                                   # mm on the sensor
    assert spread(1000.0) > 20 * in_focus and spread(2500.0) > 20 * in_focus
    orc.orc_lens_destroy(lens)


def test_setup_filter_region_quirk():
    p = camera.default_params()
    camera.setup_filter(p, 1920, 1080)
    assert (p.xres, p.yres) == (1921, 1081)          # src/lentil.h:1069-1080
    camera.setup_filter(p, 1920, 1080, region=(100, 50, 299, 149))
    assert (p.xres, p.yres, p.region_min_x, p.region_min_y) == (200, 100, 100, 50)


def test_thinlens_setup_values():
    p = camera.setup_thinlens(camera.default_params(), focal_length=35.0, fstop=1.4)
    assert abs(p.aperture_radius - (35.0 / (2.0 * float(np.float32(1.4)))) / 10.0) < 1e-12
