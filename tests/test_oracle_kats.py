"""Pins the oracle to the reference's own data (CPU only).

1. src/global.h known answers (tea<8>, rng, xor128) -- values produced from the reference header,
   recorded in SURVEY.md section 8(a) row a6.
2. tests/aperture_sampling_debug/writout.txt: the per-iteration trace of the generated
   lt_sample_aperture Newton loop (fixture tests/golden/writout_kats.json, extracted by
   tools/make_writout_kats.py).  It pins the 2x2 inverse, both update steps (aperture step undamped,
   outer-pupil step damped by 0.72), sphereToCs, csToSphere, normalise, the error-flag rules and the
   1e-8 tolerance.  The log prints 6 decimals, hence the tolerances below.
3. Structural checks of the rest of the restatement (lens_ipow, disk samplers, bokeh tables,
   draw-count formula) against independent closed forms.
The polynomial *values* of the trace cannot be checked: the generated coefficients are not in the
reference tree ("parity unpinned", see oracle/lentil_oracle.cpp header).
"""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

import common
import oracle_lib
from oracle_lib import darr

GOLD = os.path.join(common.ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def kats():
    with open(os.path.join(GOLD, "writout_kats.json")) as f:
        return json.load(f)


# ---------------------------------------------------------------------------------------------------
# 1. src/global.h
# ---------------------------------------------------------------------------------------------------
def test_tea8_known_answers(orc):
    assert orc.orc_tea8(0, 0) == 4224205021
    assert orc.orc_tea8(1, 0) == 2558915972
    assert orc.orc_tea8(12345, 7) == 3188386998
    assert orc.orc_tea8(5100, 3) == 1902035637


def test_rng_known_answers(orc):
    s = C.c_uint32(orc.orc_tea8(5100, 3))
    got = [orc.orc_rng(C.byref(s)) for _ in range(4)]
    exp = [0.499825478, 0.436503351, 0.173282743, 0.890361369]
    assert np.allclose(got, exp, rtol=0, atol=5e-10)
    # and exactly: 24-bit integers / 2^24
    for g in got:
        assert g * 16777216.0 == int(g * 16777216.0)


def test_xor128_known_answers(orc):
    st = (C.c_uint32 * 4)()
    orc.orc_xor128_init(st)
    assert [orc.orc_xor128(st) for _ in range(3)] == [3701687786, 458299110, 2500872618]


# ---------------------------------------------------------------------------------------------------
# 2. writout.txt
# ---------------------------------------------------------------------------------------------------
TOL = 2e-6     # 6 printed decimals (+ propagated print rounding where a product is checked)


def _iters(kats):
    for case in kats["cases"]:
        for it in case["iterations"]:
            yield case, it


def test_writout_header_consistency(kats):
    assert kats["header"]["aperture_rad"] == pytest.approx(6.172940)
    # "aperture = 0.5*aperturerad": the first iteration's delta_ap is the aperture point itself
    it0 = kats["cases"][0]["iterations"][0]
    assert it0["delta_ap"][0] == pytest.approx(0.5 * 6.172940, abs=1e-6)


def _scene(case):
    """Scene point of a logged case = iteration 0's `view`, which is scene - pred_out_cs_pos with
    pred_out_cs_pos = 0 (every polynomial vanishes at the zero state).  The hand-written
    "cam space pos" annotations are not used: the second one (275) contradicts its own log (285)."""
    it0 = case["iterations"][0]
    assert it0["pred_out_cs_pos"] == [0.0, 0.0, 0.0]
    return list(it0["view"])


def _curvature_radius(case):
    # the pre-loop print is scene + (0, 0, lens_outer_pupil_curvature_radius)
    return case["pre_view"][2] - _scene(case)[2]


def test_writout_inverse_2x2(orc, kats):
    """invApJ / invJ from dx1_domega0 / domega2_dx0; the log prints [0][0] [1][1] [0][1] [1][0]."""
    n = 0
    for case, it in _iters(kats):
        for jkey, ikey, dkey in (("dx1_domaga0", "invApJ", "invdetap"), ("domega2_dx0", "invJ", "invdet")):
            J = it[jkey]
            Jm = (C.c_double * 4)(J[0], J[1], J[2], J[3])
            inv = (C.c_double * 4)()
            det = C.c_double()
            orc.orc_inv2x2(Jm, inv, C.byref(det))
            got = [inv[0], inv[3], inv[1], inv[2]]
            # domega2_dx0 entries are ~1e-2 printed with 6 decimals (4-5 significant digits)
            rel, ab = (1e-3, 2e-2) if jkey == "domega2_dx0" else (2e-6, 2e-6)
            for g, e in zip(got, it[ikey]):
                assert g == pytest.approx(e, rel=rel, abs=ab), (it["k"], jkey)
            n += 1
    assert n == 80


def test_writout_aperture_step_is_undamped(orc, kats):
    """dx,dy += invApJ * delta_ap (no damping), checked with the oracle's step function."""
    for case, it in _iters(kats):
        a = it["invApJ"]
        inv = (C.c_double * 4)(a[0], a[2], a[3], a[1])          # back to [[00,01],[10,11]]
        d = darr(*it["delta_ap"])
        dx, dy = C.c_double(it["begin_dx"]), C.c_double(it["begin_dy"])
        orc.orc_newton_step(inv, d, 1.0, C.byref(dx), C.byref(dy))
        assert dx.value == pytest.approx(it["dx"], abs=3e-6)
        assert dy.value == pytest.approx(it["dy"], abs=3e-6)


def test_writout_pupil_step_is_damped_by_0_72(orc, kats):
    n = 0
    for case, it in _iters(kats):
        a = it["invJ"]
        inv = (C.c_double * 4)(a[0], a[2], a[3], a[1])
        d = darr(*it["delta_out"])
        x, y = C.c_double(it["begin_x"]), C.c_double(it["begin_y"])
        orc.orc_newton_step(inv, d, 0.72, C.byref(x), C.byref(y))
        # delta_out is printed with 6 decimals and multiplied by |invJ| ~ 100
        tol = 0.72 * (abs(a[0]) + abs(a[1]) + abs(a[2]) + abs(a[3])) * 1e-6 + 2e-6
        assert x.value == pytest.approx(it["x"], abs=tol)
        assert y.value == pytest.approx(it["y"], abs=tol)
        # and undamped / other dampings do NOT reproduce the log once the step is non-trivial
        if abs(it["y"] - it["begin_y"]) > 1e-2:
            y1 = C.c_double(it["begin_y"]); x1 = C.c_double(it["begin_x"])
            orc.orc_newton_step(inv, d, 1.0, C.byref(x1), C.byref(y1))
            assert abs(y1.value - it["y"]) > 10 * tol
            n += 1
    assert n >= 5


def test_writout_sphere_to_cs(orc, kats):
    for case, it in _iters(kats):
        R = _curvature_radius(case)
        out = it["out"]
        pos, dr = (C.c_double * 3)(), (C.c_double * 3)()
        orc.orc_sphereToCs(darr(out[0], out[1]), darr(out[2], out[3]), pos, dr, -R, R)
        assert list(pos) == pytest.approx(it["pred_out_cs_pos"], abs=TOL)
        assert list(dr) == pytest.approx(it["pred_out_cs_dir"], abs=TOL)


def test_writout_view_and_cs_to_sphere(orc, kats):
    for case, it in _iters(kats):
        R = _curvature_radius(case)
        scene = _scene(case)
        pos = it["pred_out_cs_pos"]
        view = [scene[i] - pos[i] for i in range(3)]
        assert view == pytest.approx(it["view"], abs=2e-6)
        v = darr(*view)
        orc.orc_normalise(v)
        assert list(v) == pytest.approx(it["view_normalized"], abs=TOL)
        op, od = (C.c_double * 2)(), (C.c_double * 2)()
        orc.orc_csToSphere(darr(*pos), v, op, od, -R, R)
        assert list(op) == pytest.approx(it["out_new_pos"], abs=TOL)
        assert list(od) == pytest.approx(it["out_new_dir"], abs=3e-6)
        dl = [od[0] - it["out"][2], od[1] - it["out"][3]]
        assert dl == pytest.approx(it["delta_out"], abs=4e-6)
        assert dl[0] ** 2 + dl[1] ** 2 == pytest.approx(it["sqr_err"], abs=2e-6)


def test_writout_error_flags_and_termination(orc, kats):
    """error |= 1 iff sqr_err grew, |= 2 iff sqr_ap_err grew; bits are cleared while k < 10; the loop
    stops once both squared errors are <= 1e-8 (converged case ends right after k = 10)."""
    for case in kats["cases"]:
        prev_e, prev_a = 1e30, 1e30
        for it in case["iterations"]:
            bits = orc.orc_newton_error_bits(it["sqr_err"], prev_e, it["sqr_ap_err"], prev_a,
                                             it["out"][0], it["out"][2], it["out"][3])
            logged = 0
            for b in it["errors"]:
                logged |= b
            # the printed squared errors carry 6 decimals: only compare when the change is resolvable
            if abs(it["sqr_err"] - prev_e) > 2e-6 and abs(it["sqr_ap_err"] - prev_a) > 2e-6:
                assert bits & 3 == logged & 3, (case["label"], it["k"])
            assert it["reset"] == (it["k"] < 10)
            prev_e, prev_a = it["sqr_err"], it["sqr_ap_err"]
    conv = kats["cases"][0]["iterations"]
    assert conv[-1]["k"] == 10
    last = conv[-1]
    d_ap = last["delta_ap"]
    assert d_ap[0] ** 2 + d_ap[1] ** 2 < 1e-8 and last["delta_out"][0] ** 2 + last["delta_out"][1] ** 2 < 1e-8
    # one iteration earlier the aperture error was still above the tolerance
    prev = conv[-2]["delta_ap"]
    assert prev[0] ** 2 + prev[1] ** 2 > 1e-8


# ---------------------------------------------------------------------------------------------------
# 3. structure of the remaining restatement
# ---------------------------------------------------------------------------------------------------
def test_lens_ipow_recursion(orc):
    x = 1.0000001234
    for e in range(0, 16):
        def ipow(v, n):
            if n == 0:
                return 1.0
            if n == 1:
                return v
            if n == 2:
                return v * v
            p2 = ipow(v, n // 2)
            return v * p2 * p2 if n & 1 else p2 * p2
        assert orc.orc_lens_ipow(x, e) == ipow(x, e)


def test_fast_trig_matches_formula(orc):
    pi32 = np.float32(3.14159265358979323846)
    for v in np.linspace(-6, 6, 97, dtype=np.float32):
        x = np.float32(math.fmod(float(np.float32(v + pi32)), float(np.float32(pi32 * np.float32(2))))) - pi32
        B = np.float32(4.0) / pi32
        Cc = np.float32(-4.0) / (pi32 * pi32)
        y = B * x + Cc * x * np.abs(x)
        exp = np.float32(0.225) * (y * np.abs(y) - y) + y
        assert orc.orc_fast_sin(float(v)) == float(exp)
    assert abs(orc.orc_fast_sin(0.5) - math.sin(0.5)) < 2e-3
    assert abs(orc.orc_fast_cos(0.5) - math.cos(0.5)) < 2e-3


def test_concentric_disk_sample_maps_into_unit_disk(orc):
    out = (C.c_double * 2)()
    rng = np.random.default_rng(0)
    for ox, oy in rng.random((2000, 2)):
        orc.orc_concentric_disk_sample(float(ox), float(oy), out)
        assert out[0] ** 2 + out[1] ** 2 <= 1.0 + 2e-3
    orc.orc_concentric_disk_sample(0.75, 0.5, out)       # a = 0.5, b = 0: r = 0.5, phi = 0
    assert out[0] == pytest.approx(0.5, abs=1e-3) and abs(out[1]) < 1e-3


def test_gcc_argument_order_of_the_aperture_draw(orc):
    """concentric_disk_sample(rng(seed), rng(seed), ...): GCC evaluates right to left, so ox is the
    SECOND draw and oy the first."""
    p = common.po_setup(64, 48)[0]
    seed = C.c_uint32(orc.orc_tea8(77, 5))
    d1 = orc.orc_rng(C.byref(seed))
    d2 = orc.orc_rng(C.byref(seed))
    exp = (C.c_double * 2)()
    orc.orc_concentric_disk_sample(d2, d1, exp)
    got = (C.c_double * 2)()
    orc.orc_po_aperture_sample(C.byref(p), None, 77, 5, got)
    assert got[0] == exp[0] * p.aperture_radius and got[1] == exp[1] * p.aperture_radius


def test_bokeh_tables_properties(orc):
    tex = np.load(os.path.join(GOLD, "example_bokeh_kernel_u8.npy")).astype(np.float32) / np.float32(255)
    y, x, nch = tex.shape
    B = orc.orc_bokeh_create(tex.ctypes.data, x, y, nch)
    assert B
    cdfRow = np.empty(y, np.float32); rowIdx = np.empty(y, np.int32)
    cdfCol = np.empty(x * y, np.float32); colIdx = np.empty(x * y, np.int32)
    orc.orc_bokeh_tables(B, cdfRow.ctypes.data, rowIdx.ctypes.data, cdfCol.ctypes.data, colIdx.ctypes.data)
    assert sorted(rowIdx.tolist()) == list(range(y))
    assert np.all(np.diff(cdfRow) >= 0) and cdfRow[-1] == pytest.approx(1.0, abs=1e-4)
    lum = tex[..., 0] * np.float32(0.3) + tex[..., 1] * np.float32(0.59) + tex[..., 2] * np.float32(0.11)
    rows = lum.sum(1)
    assert np.all(np.diff(rows[rowIdx]) <= 1e-3)            # rows sorted by descending weight
    cols = colIdx.reshape(y, x)
    for r in (0, 57, 249):
        assert sorted((cols[r] - r * x).tolist()) == list(range(x))
        assert np.all(np.diff(cdfCol.reshape(y, x)[r]) >= 0)
        assert np.all(np.diff(lum[r][cols[r] - r * x]) <= 1e-6)
    # sampling: u -> bright pixels more often than dark ones, output in [-1, 1]
    lens = (C.c_double * 2)()
    rng = np.random.default_rng(1)
    hit = []
    for u, v in rng.random((4000, 2)).astype(np.float32):
        orc.orc_bokeh_sample(B, float(u), float(v), lens)
        assert -1.0 <= lens[0] <= 1.0 and -1.0 <= lens[1] <= 1.0
        col = int(round(lens[0] * x / 2.0)) + (y - 1) // 2
        row = -int(round(lens[1] * y / 2.0)) + (x - 1) // 2
        hit.append(lum[row, col])
    assert np.mean(hit) > 1.5 * lum.mean()
    orc.orc_bokeh_destroy(B)


def test_host_bokeh_tables_equal_oracle(orc):
    """product host code (liblentil_host.so) vs oracle: identical tables incl. std::sort tie order."""
    from pota_amd import bokeh
    tex = np.load(os.path.join(GOLD, "example_bokeh_kernel_u8.npy")).astype(np.float32) / np.float32(255)
    y, x, nch = tex.shape
    t = bokeh.build_tables(tex)
    B = orc.orc_bokeh_create(tex.ctypes.data, x, y, nch)
    cdfRow = np.empty(y, np.float32); rowIdx = np.empty(y, np.int32)
    cdfCol = np.empty(x * y, np.float32); colIdx = np.empty(x * y, np.int32)
    orc.orc_bokeh_tables(B, cdfRow.ctypes.data, rowIdx.ctypes.data, cdfCol.ctypes.data, colIdx.ctypes.data)
    orc.orc_bokeh_destroy(B)
    assert np.array_equal(t["cdfRow"], cdfRow) and np.array_equal(t["rowIndices"], rowIdx)
    assert np.array_equal(t["cdfColumn"], cdfCol) and np.array_equal(t["columnIndices"], colIdx)


def test_draw_count_formula(orc):
    """src/lentil_filter.cpp:177-202: clamp(ceil((coc*yres)^2 * lum_mult^2 * 1e-5 * inv_density), 4, 2000)."""
    p = common.po_setup(1920, 1080)[0]
    for lum, coc in [(84.1589, 1.25), (0.5, 0.5), (3.0, 10.0), (20.0, 0.41), (1e-3, 3.0)]:
        lm = np.float32(max(0.0, math.pow(float(min(np.float32(lum), np.float32(20.0))), 0.5) * p.bidir_sample_mult))
        cy = np.float32(np.float32(coc) * np.float32(p.yres))
        csp = np.float32(float(cy) * float(cy) * (float(lm) * float(lm)) * 0.00001)
        exp = int(min(max(math.ceil(float(np.float32(csp * np.float32(p.inverse_sample_density)))), 4), 2000))
        assert orc.orc_draw_count(C.byref(p), lum, coc, p.inverse_sample_density) == exp
    p.samples_override = 256
    assert orc.orc_draw_count(C.byref(p), 1.0, 1.0, p.inverse_sample_density) == 256


def test_pow_half_vs_sqrt_after_float_narrowing():
    """The GPU evaluates std::pow(x, 0.5) (src/lentil_filter.cpp:177) as sqrt(x).  glibc's pow is not
    correctly rounded (it differs from sqrt in ~0.08 % of float-valued doubles), but the value is
    multiplied by bidir_sample_mult and narrowed to float before use -- there the two agree."""
    rng = np.random.default_rng(5)
    x = (rng.random(200000) * 20.0).astype(np.float32).astype(np.float64)
    p = np.array([math.pow(v, 0.5) for v in x])
    q = np.sqrt(x)
    for mult in (1, 3, 5, 10):
        assert np.array_equal((p * mult).astype(np.float32), (q * mult).astype(np.float32))


def test_inverse_sample_density_and_sticky_disable(orc):
    ok = C.c_int()
    d = orc.orc_inverse_sample_density(9, 1.0, 3, C.byref(ok))
    assert d == pytest.approx(1 / 9.0) and ok.value == 1
    d = orc.orc_inverse_sample_density(36, 1.5, 4, C.byref(ok))
    assert d == pytest.approx(1 / 16.0) and ok.value == 1
    orc.orc_inverse_sample_density(4, 1.0, 3, C.byref(ok))          # AA 2 != 3
    assert ok.value == 0
    orc.orc_inverse_sample_density(4, 1.0, 2, C.byref(ok))          # AA < 3
    assert ok.value == 0


def test_newton_solver_converges_on_shipped_lens(orc):
    p, model, table, keep = common.po_setup(64, 48)
    lens = orc.orc_lens_create(C.byref(table))
    it = C.c_int()
    s5, o5 = (C.c_double * 5)(), (C.c_double * 5)(0, 0, 0, 0, 0.55)
    T = orc.orc_lt_sample_aperture(lens, darr(0, -2300, 9999), darr(3.0, 3.0), s5, o5, 0.55, C.byref(it))
    assert 0.3 < T < 1.0 and 10 <= it.value <= 40
    # the solution reproduces the aperture point through the forward aperture polynomial
    v = darr(s5[0], s5[1], s5[2], s5[3], 0.55)
    assert orc.orc_poly_eval(lens, 5, v) == pytest.approx(3.0, abs=2e-4)
    assert orc.orc_poly_eval(lens, 6, v) == pytest.approx(3.0, abs=2e-4)
    orc.orc_lens_destroy(lens)


def test_seed_depends_on_attempt_plus_try_only(orc):
    """SURVEY appendix C.4 -- the identity the GPU's solve-once pipeline rests on: try t of attempt n
    uses the seed of try 0 of attempt n+t, so both traces are the same computation."""
    p, model, table, keep = common.po_setup(64, 48)
    a = (C.c_double * 2)(); b = (C.c_double * 2)()
    for n, t in [(0, 3), (17, 1), (100, 15)]:
        orc.orc_po_aperture_sample(C.byref(p), None, 1234, n + t, a)
        orc.orc_po_aperture_sample(C.byref(p), None, 1234, (n + t) + 0, b)
        assert list(a) == list(b)


def test_thinlens_chromatic_draws_one_xor128_value_per_surviving_attempt(orc):
    """Thin lens, abb_chromatic > 0 (src/lentil_filter.cpp:393-406): without optical vignetting every attempt draws
    exactly one channel from xor128 (src/global.h:22-27), so the generator has advanced by the number of attempts; the
    channels come out in thirds; channel c feeds colour component c only."""
    import ctypes as C
    import common
    import oracle_lib
    W, H, M = 48, 32, 9
    p = common.tl_setup(W, H, samples_override=32, abb_chromatic=0.7)
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.03)
    ref = oracle_lib.Frame(orc, p, n_aovs=1, keep_log=True)
    ref.run(None, None, visits)
    rc = ref.counters()
    assert rc.attempted_draws > 1000
    st = (C.c_uint32 * 4)()
    orc.orc_frame_get_xor128(ref.h, st)
    replay = (C.c_uint32 * 4)()
    orc.orc_xor128_init(replay)
    for _ in range(int(rc.attempted_draws)):
        orc.orc_xor128(replay)
    assert list(replay) == list(st)
    chan = ref.log()[:, 1] >> 30
    counts = np.bincount(chan, minlength=3)
    assert counts.min() > 0.25 * counts.sum()
    # a highlight's energy arrives in one component per draw: the three components of the sum differ pixel by pixel
    buf = ref.buffer(0)
    lit = buf[:, :3].max(axis=1) > 10.0
    assert lit.any() and (buf[lit, 0] != buf[lit, 1]).any()
    ref.close()


def test_threaded_oracle_equals_the_single_thread_one(orc):
    """tests/common.py::ThreadedOracle (the checker of the full-size GPU tests): same accepted draws and counters as one
    thread walking the stream; pixels no draw lands on bit-equal; the rest equal to fp32 summation order; fp64 shadows
    equal to rounding of the merge."""
    import common
    W, H, M = 48, 40, 9
    p, model, table, keep = common.po_setup(W, H, samples_override=32)
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.01)
    one = common.run_oracle(orc, p, table, visits, threads=1)
    thr = common.ThreadedOracle(orc, p, table, visits, 5)
    a, b = one.counters(), thr.counters()
    assert (a.visits, a.redistributed_visits, a.attempted_draws, a.accepted_draws) == (
        b.visits, b.redistributed_visits, b.attempted_draws, b.accepted_draws)
    assert a.redistributed_visits > 50
    assert np.array_equal(common.sort_log(one.log()), common.sort_log(thr.log()))
    touched = np.zeros(p.xres * p.yres, bool)
    touched[one.log()[:, 2]] = True
    assert touched.any() and not touched.all()
    assert np.array_equal(one.buffer(0)[~touched], thr.buffer(0)[~touched])
    assert np.array_equal(one.weight()[~touched], thr.weight()[~touched])
    assert np.allclose(one.buffer(0), thr.buffer(0), rtol=8e-6, atol=0)       # (fp32 sums of the same terms in another order)
    assert np.allclose(one.buffer64(0), thr.buffer64(0), rtol=1e-12, atol=0)
    assert np.allclose(one.weight64(), thr.weight64(), rtol=1e-13, atol=0)
    thr.close()


def test_threaded_oracle_merges_closest_aovs_in_stream_order(orc):
    """orc_frame_merge with closest-filtered AOVs among the frame's: the threads' survivors merged in stream order leave
    what one thread walking the stream leaves (src/lentil.h:832-837 over nonzero depths: the last candidate of the smallest
    |Z|), bit for bit -- values, z-buffer and the visit that wrote it."""
    import common
    import oracle_lib
    W, H, M = 48, 40, 9
    kinds = [0, 1, 0, 1]
    p, model, table, keep = common.po_setup(W, H, samples_override=32)
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.01, n_extra=3)
    one = common.run_oracle(orc, p, table, visits, n_aovs=4, kinds=kinds, threads=1)
    for n_threads in (2, 5, 7):
        thr = common.ThreadedOracle(orc, p, table, visits, n_threads, n_aovs=4, kinds=kinds)
        assert np.array_equal(common.sort_log(one.log()), common.sort_log(thr.log()))
        for a in (1, 3):
            assert np.array_equal(one.buffer(a), thr.buffer(a)), (n_threads, a)
            assert np.array_equal(one.resolve(a), thr.resolve(a)), (n_threads, a)
        assert np.array_equal(one.zbuffer(), thr.zbuffer())
        assert np.array_equal(one.zvisit(), thr.zvisit())
        for a in (0, 2):
            # (another summation order -- a pixel's own visits first, then the draws in visit order: both sides are fp32 sums of
            # the same terms, each within a few 1e-6 of the exact one the shadows hold)
            assert np.allclose(one.buffer(a), thr.buffer(a), rtol=8e-6, atol=0)
            ex = one.buffer64(a)
            m = ex != 0
            assert float(np.max(np.abs(thr.buffer(a)[m] - ex[m]) / np.abs(ex[m]))) < 1e-5
        # (draws carry candidates across thread boundaries: some pixel's survivor comes from another thread's range)
        thr.close()
    assert (one.zvisit() != 0xFFFFFFFF).sum() >= W * H
    one.close()


def test_oracle_camera_motion_interpolates_the_keys(orc):
    """orc_frame_set_camera_motion: a visit at lentil_time t sees ((b - a) * f) + a of the two keys around t; two equal
    keys are the static camera; times outside the shutter clamp."""
    import common
    import oracle_lib
    W, H, M = 32, 24, 9
    p, model, table, keep = common.po_setup(W, H, samples_override=16)
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.03)
    lens = orc.orc_lens_create(C.byref(table))

    def run(keys, times, shutter=None):
        cols["raydir_time"][:, 3] = times
        f = oracle_lib.Frame(orc, p, n_aovs=1, keep_log=True)
        if keys is not None:
            f.set_camera_motion(keys)
        if shutter is not None:
            f.set_camera_shutter(*shutter)
        f.run(lens, None, visits)
        log = common.sort_log(f.log())
        f.close()
        return log
    ident = np.eye(4, dtype=np.float32)
    moved = ident.copy(); moved[3, 0] = 20.0
    n = visits.n
    still = run(None, np.zeros(n, np.float32))
    assert still.shape[0] > 100
    assert np.array_equal(run(np.stack([ident, ident]), np.full(n, 0.37, np.float32)), still)
    at_end = run(np.stack([ident, moved]), np.ones(n, np.float32))
    assert not np.array_equal(at_end, still)
    assert np.array_equal(run(np.stack([ident, moved]), np.full(n, 7.0, np.float32)), at_end)        # clamped to 1
    assert np.array_equal(run(np.stack([ident, moved]), np.full(n, -3.0, np.float32)), still)        # clamped to 0
    half = moved.copy(); half[3, 0] = 10.0
    assert np.array_equal(run(np.stack([ident, moved]), np.full(n, 0.5, np.float32)), run(np.stack([half, half]), np.zeros(n, np.float32)))
    assert np.array_equal(run(np.stack([ident, half, moved]), np.full(n, 0.5, np.float32)), run(np.stack([half, half]), np.zeros(n, np.float32)))
    # lentil_time is Arnold's absolute sample time and the keys span the camera's shutter (orc_frame_set_camera_shutter): a centred
    # shutter -0.25 ... 0.25 has the first key at -0.25, the last at 0.25 and their mean at 0
    centred = (-0.25, 0.25)
    assert np.array_equal(run(np.stack([ident, moved]), np.full(n, -0.25, np.float32), centred), still)
    assert np.array_equal(run(np.stack([ident, moved]), np.full(n, 0.25, np.float32), centred), at_end)
    assert np.array_equal(run(np.stack([ident, moved]), np.full(n, 0.9, np.float32), centred), at_end)        # beyond the shutter: last key
    assert np.array_equal(run(np.stack([ident, moved]), np.zeros(n, np.float32), centred), run(np.stack([half, half]), np.zeros(n, np.float32)))
    assert not np.array_equal(run(np.stack([ident, moved]), np.zeros(n, np.float32), centred), still)         # (under 0 ... 1 time 0 is key 0)
    orc.orc_lens_destroy(lens)
