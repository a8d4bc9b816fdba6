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


# ---------------------------------------------------------------------------------------------------
# liblentil_host.so (product, C++) against the oracle -- SURVEY section 8f ranks 1 and 2
# ---------------------------------------------------------------------------------------------------
import pytest
from pota_amd import hostlib, _abi


@pytest.fixture(scope="module", params=["double_gauss_50mm", "petzval_58mm"])
def both(request, orc):
    model = camera.LensModel(request.param)
    hl = hostlib.HostLens(model.spec)
    table, keep = lens_io.make_lens_table(model.spec)
    ol = orc.orc_lens_create(C.byref(table))
    yield model, hl, ol
    orc.orc_lens_destroy(ol)
    hl.close()


def test_host_lens_primitives_bitwise(orc, both):
    model, hl, ol = both
    lib = hostlib.load()
    rng = np.random.default_rng(3)
    a5, b5 = (C.c_double * 5)(), (C.c_double * 5)()
    for _ in range(100):
        v = [rng.uniform(-12, 12), rng.uniform(-12, 12), rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), 0.55]
        assert lib.lentil_host_lens_evaluate(hl.h, darr(*v), a5) == orc.orc_lens_evaluate(ol, darr(*v), b5)
        assert list(a5)[:4] == list(b5)[:4]
        i1, o1 = darr(v[0], v[1], 0, 0, 0.55), darr(rng.uniform(-3, 3), rng.uniform(-3, 3), 0, 0, 0)
        i2, o2 = darr(*i1), darr(*o1)
        dist = rng.uniform(0, 4)
        lib.lentil_host_lens_pt_sample_aperture(hl.h, i1, o1, dist)
        orc.orc_pt_sample_aperture(ol, i2, o2, dist)
        assert list(i1) == list(i2) and list(o1) == list(o2)
        scene = darr(rng.uniform(-2000, 2000), rng.uniform(-1500, 1500), rng.uniform(500, 9000))
        ap = darr(rng.uniform(-4, 4), rng.uniform(-4, 4))
        s1, q1, s2, q2 = (C.c_double * 5)(), (C.c_double * 5)(0, 0, 0, 0, .55), (C.c_double * 5)(), (C.c_double * 5)(0, 0, 0, 0, .55)
        t1 = lib.lentil_host_lens_lt_sample_aperture(hl.h, scene, ap, s1, q1, 0.55)
        t2 = orc.orc_lt_sample_aperture(ol, scene, ap, s2, q2, 0.55, None)
        assert t1 == t2 and np.array_equal(np.array(s1), np.array(s2), equal_nan=True)
        assert np.array_equal(np.array(q1), np.array(q2), equal_nan=True)


def test_host_setup_functions_bitwise(orc, both):
    model, hl, ol = both
    lib = hostlib.load()
    lam = float(np.float32(550.0)) * 0.001
    for s in (-3.0, 0.0, 1.7, 6.0):
        assert lib.lentil_host_camera_get_y0_intersection_distance(hl.h, s, lam) == \
            orc.orc_camera_get_y0_intersection_distance(ol, s, lam)
    shift_h = lib.lentil_host_logarithmic_focus_search(hl.h, 1500.0, lam)
    assert shift_h == orc.orc_logarithmic_focus_search(ol, 1500.0, lam)
    assert shift_h == model.logarithmic_focus_search(1500.0, lam)           # numpy mirror
    f1, r1, f2, r2 = C.c_double(), C.c_double(), C.c_double(), C.c_double()
    for target in (2.8, 5.6, 11.0):
        lib.lentil_host_trace_backwards_for_fstop(hl.h, target, lam, C.byref(f1), C.byref(r1))
        orc.orc_trace_backwards_for_fstop(ol, target, lam, C.byref(f2), C.byref(r2))
        assert (f1.value, r1.value) == (f2.value, r2.value)
        if f1.value:
            assert f1.value >= target and 0 < r1.value <= model.k["lens_outer_pupil_radius"]
    d1, d2 = C.c_double(), C.c_double()
    ok1 = lib.lentil_host_trace_ray_focus_check(hl.h, shift_h, lam, C.byref(d1))
    ok2 = orc.orc_trace_ray_focus_check(ol, shift_h, lam, C.byref(d2))
    assert ok1 == ok2 == 1 and d1.value == d2.value
    assert abs(d1.value - 1500.0) < 25.0        # the focus test ray crosses the axis at the focus distance


def test_setup_with_fstop_stops_down(both):
    model, hl, ol = both
    p = camera.default_params()
    p2, _ = camera.setup_po(p, model, focus_dist=150.0, fstop=8.0)
    assert 0 < p2.aperture_radius < model.k["lens_aperture_radius_at_fstop"]
    p3, _ = camera.setup_po(camera.default_params(), model, focus_dist=150.0)
    assert p3.aperture_radius == model.k["lens_aperture_radius_at_fstop"]


def _fw_po_pair(orc, p, hl, ol, sx, sy, r1, r2, deriv, tables=None, ob=None):
    lib = hostlib.load()
    lam = float(np.float32(550.0)) * 0.001
    res = []
    for which in (0, 1):
        st = (C.c_uint32 * 4)()
        lib.lentil_host_xor128_init(st)
        o, d, w = (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 3)(1, 1, 1)
        a, b, tr = C.c_double(r1), C.c_double(r2), C.c_int()
        if which == 0:
            lib.lentil_host_trace_ray_fw_po(C.byref(p), hl.h, tables, st, lam, sx, sy, C.byref(a), C.byref(b), deriv, o, d, w, C.byref(tr))
        else:
            orc.orc_trace_ray_fw_po(C.byref(p), ol, ob, st, lam, sx, sy, C.byref(a), C.byref(b), deriv, o, d, w, C.byref(tr))
        res.append((list(o), list(d), list(w), tr.value, a.value, b.value, list(st)))
    return res


def test_forward_po_rays_bitwise(orc, both):
    """Camera::trace_ray_fw_po (src/lentil.h:283-427): host library vs oracle, incl. the xor128 retries."""
    model, hl, ol = both
    p, _ = camera.setup_po(camera.setup_filter(camera.default_params(), 640, 360), model)
    rng = np.random.default_rng(9)
    n_retry = 0
    for _ in range(300):
        sx, sy = rng.uniform(-1.1, 1.1), rng.uniform(-0.7, 0.7)
        h, o = _fw_po_pair(orc, p, hl, ol, sx, sy, rng.random(), rng.random(), 0)
        assert np.array_equal(np.array(h[0]), np.array(o[0]), equal_nan=True)
        assert np.array_equal(np.array(h[1]), np.array(o[1]), equal_nan=True)
        assert h[2:] == o[2:]
        n_retry += h[3] > 0
        if h[2] != [0.0, 0.0, 0.0]:
            assert abs(np.linalg.norm(h[1]) - 1) < 1e-5 and h[1][2] < 0      # unit direction, looking down -z
    assert n_retry > 0          # the vignetting-retry path was exercised


@pytest.mark.parametrize("coma", [0.0, 0.6])
def test_forward_thinlens_rays_bitwise(orc, coma):
    lib = hostlib.load()
    p = camera.setup_thinlens(camera.setup_filter(camera.default_params(), 640, 360))
    p.optical_vignetting_distance = 2.0        # exercises retries
    p.abb_coma = coma                          # src/lentil.h:490-491
    rng = np.random.default_rng(10)
    n_retry = 0
    for _ in range(300):
        sx, sy, r1, r2 = rng.uniform(-1, 1), rng.uniform(-0.6, 0.6), rng.random(), rng.random()
        res = []
        for which in (0, 1):
            st = (C.c_uint32 * 4)()
            lib.lentil_host_xor128_init(st)
            o, d, w = (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 3)(1, 1, 1)
            a, b, tr = C.c_double(r1), C.c_double(r2), C.c_int()
            if which == 0:
                lib.lentil_host_trace_ray_fw_thinlens(C.byref(p), None, st, sx, sy, C.byref(a), C.byref(b), 0, o, d, w, C.byref(tr))
            else:
                orc.orc_trace_ray_fw_thinlens(C.byref(p), None, st, sx, sy, C.byref(a), C.byref(b), 0, o, d, w, C.byref(tr))
            res.append((list(o), list(d), list(w), tr.value, list(st)))
        assert res[0] == res[1]
        n_retry += res[0][3] > 0
    assert n_retry > 0


def test_camera_create_ray_differentials(both):
    """camera_create_ray (src/lentil_camera.cpp:78-125): finite differences with step 0.001."""
    model, hl, ol = both
    lib = hostlib.load()
    p, _ = camera.setup_po(camera.setup_filter(camera.default_params(), 640, 360), model)
    st = (C.c_uint32 * 4)()
    lib.lentil_host_xor128_init(st)
    ray = hostlib.CameraRay()
    inp = (C.c_float * 6)(0.1, -0.05, 1.0, 1.0, 0.5, 0.5)
    lib.lentil_host_camera_create_ray(C.byref(p), hl.h, None, st, 0.55, 1.0, inp, C.byref(ray))
    assert list(ray.weight) == [1.0, 1.0, 1.0]
    assert abs(np.linalg.norm(list(ray.dir)) - 1) < 1e-5
    # moving the sensor point in +x changes the ray direction in x (and hardly in y)
    assert abs(ray.dDdx[0]) > 10 * abs(ray.dDdx[1]) and abs(ray.dDdy[1]) > 10 * abs(ray.dDdy[0])
    ps = (C.c_float * 2)()
    lib.lentil_host_camera_reverse_ray(0.36, (C.c_float * 3)(10.0, 5.0, -100.0), ps)
    assert abs(ps[0] - 10.0 / 36.0) < 1e-6 and abs(ps[1] - 5.0 / 36.0) < 1e-6
