"""More GPU-vs-oracle parity cases (round 2): branches of the device code that the first round's suite never ran --
the polygonal aperture of the polynomial-optics path, enable_dof = 0, cylindrical pupils -- full-size geometry for
BASELINE configs 1 and 2, and the two captured input files the reference's own tests hold
(tests/golden/sampledata_2k.npy, po_bidir_positions.npy; tools/make_sample_fixtures.py).

Bar as everywhere: accepted-draw lists (visit, attempt, pixel) bit-identical, counters equal, frames within 1e-5.
"""
import ctypes as C
import os

import numpy as np
import pytest

import common
import oracle_lib
from pota_amd import _abi, capi, workload
from test_gpu_parity import _band_parity_at_full_geometry, check_frame, check_logs, gpu_run

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(common.ROOT, "tests", "golden")


def _same_counters(c, ref):
    rc = ref.counters()
    assert (c.redistributed_visits, c.attempted_draws, c.accepted_draws) == (
        rc.redistributed_visits, rc.attempted_draws, rc.accepted_draws)
    return rc


@pytest.mark.parametrize("blades", [3, 5, 6, 8])
def test_po_polygonal_aperture(orc, gpu_ctx_factory, blades):
    """bokeh_aperture_blades > 2 in polynomial-optics mode: Camera::lens_sample_triangular_aperture with
    radius = aperture_radius, no image option (src/lentil.h:597-609, 964-982; the threshold differs from the thin
    lens's)."""
    W, H, M = 64, 48, 9
    p, model, table, keep = common.po_setup(W, H, samples_override=48, bokeh_aperture_blades=blades)
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.02)
    ref = common.run_oracle(orc, p, table, visits)
    assert ref.counters().redistributed_visits > 100
    ctx = gpu_ctx_factory()
    for lens_mode in (0, 1):                      # compiled kernel, table interpreter
        c = gpu_run(ctx, p, table, visits, lens_mode=lens_mode)
        _same_counters(c, ref)
        check_logs(ctx, ref)
        check_frame(ctx, ref)
    # the aperture draws themselves, bit for bit
    rng = np.random.default_rng(blades)
    a = rng.integers(0, 2 ** 32, 4096, dtype=np.uint64).astype(np.uint32)
    b = rng.integers(0, 6000, 4096, dtype=np.uint64).astype(np.uint32)
    got = ctx.test_aperture_sample(a, b)
    want = np.empty_like(got)
    xy = (C.c_double * 2)()
    for i in range(a.shape[0]):
        orc.orc_po_aperture_sample(C.byref(p), None, int(a[i]), int(b[i]), xy)
        want[i] = xy[0], xy[1]
    assert np.array_equal(got, want)


def test_po_enable_dof_off(orc, gpu_ctx_factory):
    """enable_dof = 0: every backward trace goes through the aperture's centre (src/lentil.h:592-595)."""
    W, H, M = 64, 48, 9
    p, model, table, keep = common.po_setup(W, H, samples_override=32, enable_dof=0)
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.02)
    ref = common.run_oracle(orc, p, table, visits)
    assert ref.counters().redistributed_visits > 100
    ctx = gpu_ctx_factory()
    c = gpu_run(ctx, p, table, visits)
    _same_counters(c, ref)
    check_logs(ctx, ref)
    check_frame(ctx, ref)


def test_po_cylindrical_outer_pupil(orc, gpu_ctx_factory):
    """A lens whose outer pupil is a cylinder (axis along y: "cyl-y", the anamorphic front attachment of
    pota_amd/lenses/anamorphic_petzval_58mm.json): cylinderToCs / csToCylinder in every Newton iteration
    (src/lens.h:156-221, src/lentil.h:387-389).  Table interpreter (no compiled kernel is shipped for this table)."""
    W, H, M = 64, 48, 9
    p, model, table, keep = common.po_setup(W, H, lens="anamorphic_petzval_58mm", samples_override=32)
    assert table.lens_outer_pupil_geometry != 0
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.03)
    ref = common.run_oracle(orc, p, table, visits)
    rc = ref.counters()
    assert rc.redistributed_visits > 100 and rc.accepted_draws > 1000      # the solver does converge on this table
    ctx = gpu_ctx_factory()
    c = gpu_run(ctx, p, table, visits, compiled=False)
    _same_counters(c, ref)
    check_logs(ctx, ref)
    check_frame(ctx, ref)
    # lt_sample_aperture alone, fp64 bit for bit
    rng = np.random.default_rng(11)
    n = 2048
    scene = np.stack([rng.uniform(-400, 400, n), rng.uniform(-300, 300, n), rng.uniform(600, 4000, n)], 1)
    ap = rng.uniform(-0.7, 0.7, (n, 2)) * float(p.aperture_radius)
    lam = float(np.float32(0.55))
    sensor, out, T = ctx.test_lt_sample_aperture(scene, ap, lam)
    lens = orc.orc_lens_create(C.byref(table))
    es, eo, eT = np.empty_like(sensor), np.empty_like(out), np.empty_like(T)
    s5, o5 = (C.c_double * 5)(), (C.c_double * 5)()
    for i in range(n):
        for k in range(5):
            o5[k] = 0.0
        o5[4] = lam
        eT[i] = orc.orc_lt_sample_aperture(lens, oracle_lib.darr(*scene[i]), oracle_lib.darr(*ap[i]), s5, o5, lam, None)
        es[i] = list(s5)
        eo[i] = list(o5)
    orc.orc_lens_destroy(lens)
    assert (eT > 0).mean() > 0.2
    assert np.array_equal(T, eT, equal_nan=True)
    assert np.array_equal(sensor, es, equal_nan=True)
    assert np.array_equal(out, eo, equal_nan=True)


def test_config1_thinlens_512_full_frame(orc, gpu_ctx_factory):
    """BASELINE config 1 at its full size: thin lens, 512 x 512, 64 draws, beauty only -- the whole frame against the
    oracle (2.4 M visits; the thin-lens draw has no solver, the oracle takes a few seconds)."""
    W = H = 512
    M = 9
    p = common.tl_setup(W, H, samples_override=64)
    visits, cols = common.make_stream(p, W, H, M, f_hi=2.0 ** -9)
    ref = common.run_oracle(orc, p, None, visits)
    assert ref.counters().redistributed_visits > 3000
    ctx = gpu_ctx_factory()
    c = gpu_run(ctx, p, None, visits)
    _same_counters(c, ref)
    check_logs(ctx, ref)
    check_frame(ctx, ref)


def test_config2_1080p_bands(orc, gpu_ctx_factory):
    """BASELINE config 2 (double-gauss, 1920 x 1080, 256 draws, beauty): three bands of rows at full-frame geometry --
    top edge, centre, bottom edge -- against the oracle."""
    W, H, M = 1920, 1080, 9
    p, model, table, keep = common.po_setup(W, H, samples_override=256)
    ctx = gpu_ctx_factory()
    for y0 in (0, 538, 1076):
        _band_parity_at_full_geometry(orc, ctx, p, table, W, M, y0, 4, 2.0 ** -9)


def _captured_stream(p, W, H, pos_cs, rgba, depth):
    """Visits from captured camera-space positions: identity world_to_camera, pixel = pinhole projection of the
    position (ragged stream with explicit pixels), uniform inverse density."""
    n = pos_cs.shape[0]
    t = common.tan_half_fov(p)
    fx = pos_cs[:, 0] / (-pos_cs[:, 2]) / t
    fy = pos_cs[:, 1] / (-pos_cs[:, 2]) / (t * H / W)
    px = np.clip(((fx + 1.0) * 0.5 * W).astype(np.int64), 0, W - 1).astype(np.uint32)
    py = np.clip(((1.0 - fy) * 0.5 * H).astype(np.int64), 0, H - 1).astype(np.uint32)
    z = np.zeros((n, 4), np.float32)
    nrm = np.linalg.norm(pos_cs, axis=1, keepdims=True)
    cols = {
        "rgba": np.ascontiguousarray(rgba, np.float32),
        "pos_z": np.ascontiguousarray(np.concatenate([pos_cs, depth[:, None]], 1), np.float32),
        "raydir_time": np.ascontiguousarray(np.concatenate([pos_cs / nrm, np.zeros((n, 1))], 1), np.float32),
        "volume_ignore": z.copy(), "transmission": z.copy(), "extra": [],
        "pixel": (px | (py << 16)).astype(np.uint32),
    }
    visits, keep = capi.make_visits(cols, visits_per_pixel=0)
    return visits, (cols, keep)


@pytest.mark.parametrize("mode", ["thinlens", "po"])
def test_captured_light_grid_samples(orc, gpu_ctx_factory, mode):
    """The reference's captured AOV samples (tests/cuda/sampledata.txt, every 26th line: RGBA 84.1589, depth, camera-
    space position of the light-grid scene) as the input of a pass, thin lens and polynomial optics."""
    d = np.load(os.path.join(GOLDEN, "sampledata_2k.npy"))
    W, H = 96, 64
    if mode == "thinlens":
        p, table = common.tl_setup(W, H, samples_override=24), None
    else:
        p, model, table, keep = common.po_setup(W, H, samples_override=24, focus_dist=110.0)
    visits, keepv = _captured_stream(p, W, H, d[:, 5:8].astype(np.float64), d[:, 0:4], d[:, 4])
    ref = common.run_oracle(orc, p, table, visits)
    assert ref.counters().redistributed_visits > 500
    ctx = gpu_ctx_factory()
    c = gpu_run(ctx, p, table, visits)
    _same_counters(c, ref)
    check_logs(ctx, ref)
    check_frame(ctx, ref)


@pytest.mark.parametrize("mode", ["thinlens", "po"])
def test_captured_sphere_positions(orc, gpu_ctx_factory, mode):
    """The 3 699 camera-space positions of tests/po_bidir_debug/po_bidir_spheres_debug_position.txt (the debug dump of
    the reference's bidirectional spheres scene) as highlights."""
    pos = np.load(os.path.join(GOLDEN, "po_bidir_positions.npy")).astype(np.float64)
    n = pos.shape[0]
    W, H = 96, 64
    if mode == "thinlens":
        p, table = common.tl_setup(W, H, samples_override=16), None
    else:
        p, model, table, keep = common.po_setup(W, H, samples_override=16, focus_dist=90.0)
    rgba = np.tile(np.array([workload.HIGHLIGHT_RADIANCE] * 3 + [1.0], np.float32), (n, 1))
    visits, keepv = _captured_stream(p, W, H, pos, rgba, np.linalg.norm(pos, axis=1).astype(np.float32))
    ref = common.run_oracle(orc, p, table, visits)
    assert ref.counters().redistributed_visits > 1000
    ctx = gpu_ctx_factory()
    c = gpu_run(ctx, p, table, visits)
    _same_counters(c, ref)
    check_logs(ctx, ref)
    check_frame(ctx, ref)


def test_config5_8k_bands(orc, gpu_ctx_factory):
    """BASELINE config 5's geometry (double-gauss, 7680 x 4320, 2048 draws -- above the reference's clamp of 2000,
    hence an override): single rows at the top edge and in the centre of the frame against the oracle."""
    W, H, M = 7680, 4320, 9
    p, model, table, keep = common.po_setup(W, H, samples_override=2048)
    ctx = gpu_ctx_factory()
    for y0 in (0, 2160):
        _band_parity_at_full_geometry(orc, ctx, p, table, W, M, y0, 1, 2.0 ** -11)


def _moving_camera_keys(n_keys):
    """world-to-camera matrices (row-vector convention, translation in row 3) of a camera that trucks sideways and pans a
    little over the shutter: enough to move a highlight's camera-space position by a good fraction of its depth."""
    keys = []
    for k in range(n_keys):
        t = k / max(1, n_keys - 1)
        a = 0.06 * t
        m = np.eye(4, dtype=np.float32)
        m[0, 0], m[0, 2], m[2, 0], m[2, 2] = np.cos(a), -np.sin(a), np.sin(a), np.cos(a)
        m[3, 0], m[3, 1], m[3, 2] = 12.0 * t, -3.0 * t * t, 4.0 * t
        keys.append(m)
    return np.stack(keys)


@pytest.mark.gpu
@pytest.mark.parametrize("n_keys,ragged,shutter", [(2, False, None), (5, False, None), (3, True, None), (5, False, (-0.25, 0.25)),
                                                   (3, True, (0.0, 0.5))])
def test_moving_camera_per_sample_time(orc, gpu_ctx_factory, n_keys, ragged, shutter):
    """Every AOV sample is taken to camera space with the matrix of its own time -- AiWorldToCameraMatrix(camera,
    lentil_time), src/lentil_filter.cpp:141-144: matrix keys over the shutter (lentil_hip_set_camera_motion), the
    lentil_time column deciding between them per visit.  Oracle and HIP path on the same keys and times: accepted-draw
    lists bit-identical, frames within 1e-5; and the keys do matter (the static camera gives another frame)."""
    import test_gpu_parity as tp
    W, H, M = 64, 48, 9
    p, model, table, keep = common.po_setup(W, H, samples_override=48)
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.02)
    cols["raydir_time"] = cols["raydir_time"].copy()
    rng = np.random.default_rng(11)
    # lentil_time is Arnold's absolute sample time; the keys span the camera's shutter (lentil_hip_set_camera_shutter; 0 ... 1
    # unless set -- a centred shutter is -0.25 ... 0.25).  Incl. times outside the shutter (clamped to the first / last key)
    s0, s1 = shutter if shutter else (0.0, 1.0)
    cols["raydir_time"][:, 3] = rng.uniform(s0 - 0.1 * (s1 - s0), s1 + 0.1 * (s1 - s0), visits.n).astype(np.float32)
    if ragged:
        pix = np.arange(visits.n) // M
        cols["pixel"] = ((pix % W) | ((pix // W) << 16)).astype(np.uint32)
        visits, keepv = capi.make_visits(cols, visits_per_pixel=0)
    else:
        visits, keepv = capi.make_visits(cols, visits_per_pixel=M, pixels_per_row=W)
    keys = _moving_camera_keys(n_keys)
    lens = orc.orc_lens_create(C.byref(table))
    ref = oracle_lib.Frame(orc, p, n_aovs=1, keep_log=True)
    ref.set_camera_motion(keys)
    if shutter:
        ref.set_camera_shutter(*shutter)
    ref.run_auto(lens, None, visits)
    still = oracle_lib.Frame(orc, p, n_aovs=1, keep_log=True)
    still.run_auto(lens, None, visits)
    orc.orc_lens_destroy(lens)
    assert ref.counters().redistributed_visits > 100
    assert not np.array_equal(common.sort_log(ref.log()), common.sort_log(still.log()))
    ctx = gpu_ctx_factory()
    ctx.set_camera_motion(keys)
    if shutter:
        ctx.set_camera_shutter(*shutter)
    for _ in range(2):                     # second pass: blind / streamed
        c = tp.gpu_run(ctx, p, table, visits)
        rc = ref.counters()
        assert (c.redistributed_visits, c.attempted_draws, c.accepted_draws) == (rc.redistributed_visits, rc.attempted_draws, rc.accepted_draws)
        tp.check_logs(ctx, ref)
        tp.check_frame(ctx, ref)
    ctx.set_camera_motion(None)            # back to the parameters' static matrix
    tp.gpu_run(ctx, p, table, visits)
    tp.check_logs(ctx, still)
    ref.close(); still.close()


@pytest.mark.gpu
@pytest.mark.parametrize("run_len", [200, 320, 321, 1500])
def test_ragged_stream_with_a_long_same_pixel_run(orc, gpu_ctx_factory, run_len):
    """scan_runs_kernel (ragged streams: runs of one pixel's samples in iterator order): the run's first lane sums its own 64
    visits and the 256 after them in order; visits further into a run than that add themselves (kRunFollow, ADVICE round 3:
    one lane must not walk a run of thousands).  Runs up to 320 visits stay bit-exact sequential sums; longer ones are inside
    1e-5.  One pixel gets a run of `run_len` visits starting at an odd offset, the rest of the stream keeps 9 per pixel;
    an extra gaussian AOV rides along."""
    import test_gpu_parity as tp
    W, H, M = 64, 48, 9
    p, model, table, keep = common.po_setup(W, H, samples_override=48)
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.02, n_extra=1)
    n = visits.n
    pix = np.arange(n) // M
    first = 37 * M + 5                      # the long run begins mid-pixel, mid-wave
    target = pix[first]
    pix[first:first + run_len] = target
    # (the pixels the run swallowed keep their remaining visits as shorter runs; pixel `target` appears in ONE run)
    pix[first - 5:first] = target
    cols["pixel"] = ((pix % W) | ((pix // W) << 16)).astype(np.uint32)
    visits, keepv = capi.make_visits(cols, visits_per_pixel=0)
    lens = orc.orc_lens_create(C.byref(table))
    ref = oracle_lib.Frame(orc, p, n_aovs=2, keep_log=True)
    ref.run_auto(lens, None, visits)
    orc.orc_lens_destroy(lens)
    ctx = gpu_ctx_factory()
    for _ in range(2):
        c = tp.gpu_run(ctx, p, table, visits, n_aovs=2)
        rc = ref.counters()
        assert (c.redistributed_visits, c.attempted_draws, c.accepted_draws) == (rc.redistributed_visits, rc.attempted_draws, rc.accepted_draws)
        tp.check_logs(ctx, ref)
        tp.check_frame(ctx, ref, n_aovs=2)
        touched = np.zeros(p.xres * p.yres, bool)
        touched[ref.log()[:, 2]] = True
        lin = int(target % W) + int(target // W) * p.xres
        buf, w = ctx.download_accum(0)
        if not touched[lin] and run_len + 5 <= 320:
            assert np.array_equal(buf[lin], ref.buffer(0)[lin]) and w[lin] == ref.weight()[lin]        # the sequential sum, bit for bit
        others = ~touched
        others[lin] = False
        assert np.array_equal(buf[others], ref.buffer(0)[others])
    ref.close()


@pytest.mark.gpu
def test_every_context_gets_streams_that_run_together(gpu_ctx_factory):
    """lentil_hip_create picks the four streams a streamed pass keeps kernels resident on so that they do not share a
    hardware queue (the runtime multiplexes streams onto GPU_MAX_HW_QUEUES queues, 4 by default): true for the first
    context of a process by luck, for the later ones by the probe."""
    import os
    if os.environ.get("GPU_MAX_HW_QUEUES", "4").isdigit() and int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) < 4:
        pytest.skip("fewer than four hardware queues allowed")
    ctxs = [gpu_ctx_factory() for _ in range(4)]
    assert all(c.streams_concurrent() >= 1 for c in ctxs)


@pytest.mark.gpu
def test_two_contexts_pass_at_the_same_time(orc, gpu_ctx_factory):
    """Two contexts on one device, driven from two threads: a streamed pass keeps kernels resident, so only one runs per
    device at a time -- the other context does not wait for it, its pass takes the chunked form.  Every pass of both
    equals the oracle, whichever form it took."""
    import threading
    W, H, M = 96, 64, 9
    p, model, table, keep = common.po_setup(W, H, samples_override=48)
    streams = [common.make_stream(p, W, H, M, f_hi=0.004, seed=s) for s in (0x5EED, 0xBEEF)]
    refs = [common.run_oracle(orc, p, table, v) for v, _ in streams]
    ctxs = [gpu_ctx_factory() for _ in range(2)]
    errors, forms = [], [[], []]

    def work(i):
        try:
            ctx = ctxs[i]
            for k in range(6):
                c = gpu_run(ctx, p, table, streams[i][0])
                rc = refs[i].counters()
                assert (c.redistributed_visits, c.attempted_draws, c.accepted_draws) == (
                    rc.redistributed_visits, rc.attempted_draws, rc.accepted_draws)
                forms[i].append(int(c.streamed))
                check_logs(ctx, refs[i])
                check_frame(ctx, refs[i])
        except Exception as e:      # noqa: BLE001
            errors.append((i, e))

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errors, "context %d: %r" % errors[0]
    assert not any(t.is_alive() for t in th)
    assert sum(forms[0]) + sum(forms[1]) >= 1       # (the first pass of a context is never streamed)


@pytest.mark.parametrize("what", ["zero", "negative_zero", "nan", "mixed"])
@pytest.mark.parametrize("ragged,own_log", [(False, True), (True, True), (False, False)], ids=["uniform", "ragged", "no-log"])
def test_degenerate_depths_and_closest_aovs(orc, gpu_ctx_factory, what, ragged, own_log):
    """src/lentil.h:832-837 treats a z-buffer value of 0 as "empty": a sample at |Z| == 0 wins and re-opens the pixel -- the next
    sample replaces it whatever its depth --, a NaN written into an open pixel is never replaced and is ignored anywhere else:
    at a pixel that sees such a sample the reference's image depends on the ORDER of the samples there.  Round 4 refused such a
    pass; since round 5 the library replays those pixels in visit order (lentil_closest_replay.h) and the frame equals the
    single-threaded oracle's: closest-filtered AOV and lentil_debug bit for bit, gaussian AOVs as ever.  Degenerate depths on
    samples that stay in their pixel AND on redistributed ones (whose draws carry the depth to every pixel they land on),
    two bad samples in one pixel, a bad sample as a pixel's first and as its last; "mixed": zeros and NaNs side by side.
    "no-log": the caller set no draw log -- the library sets one up, wipes the frame and runs the pass again."""
    W, H, M = 48, 32, 9
    p, model, table, keep = common.po_setup(W, H, samples_override=32)
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.02, n_extra=2)
    n = cols["rgba"].shape[0]
    rng = np.random.default_rng(3)
    hi = np.nonzero(cols["rgba"][:, 0] > 2.0)[0]
    bad = np.concatenate([rng.choice(n, 14, replace=False), rng.choice(hi, min(5, hi.size), replace=False),
                          np.array([0, M - 1, 5 * M, 5 * M + 1, 77 * M + 4, 77 * M + 5, 77 * M + 6])])      # first / last of a pixel, neighbours
    vals = {"zero": [np.float32(0.0)], "negative_zero": [np.float32(-0.0)], "nan": [np.float32(np.nan)],
            "mixed": [np.float32(0.0), np.float32(np.nan), np.float32(-0.0)]}[what]
    cols["pos_z"][bad, 3] = np.array([vals[i % len(vals)] for i in range(bad.size)], np.float32)
    layout = dict(visits_per_pixel=M, pixels_per_row=W)
    if ragged:
        pix = np.arange(n, dtype=np.uint64) // M
        cols["pixel"] = ((pix % W).astype(np.uint32) | ((pix // W).astype(np.uint32) << 16)).astype(np.uint32)
        layout = dict(visits_per_pixel=0)
    kinds = [_abi.FILTER_GAUSSIAN, _abi.FILTER_CLOSEST, _abi.FILTER_CLOSEST_DEBUG]
    # (lentil_debug has no visit column; the oracle reads every column: a dummy one there, its values are ignored)
    gcols = dict(cols); gcols["extra"] = [cols["extra"][0], None]
    visits, keepv = capi.make_visits(gcols, **layout)
    ocols = dict(cols); ocols["extra"] = [cols["extra"][0], np.zeros_like(cols["rgba"])]
    ovisits, okeep = capi.make_visits(ocols, **layout)
    lens = orc.orc_lens_create(C.byref(table))
    ref = oracle_lib.Frame(orc, p, n_aovs=3, kinds=kinds, keep_log=True)
    ref.run(lens, None, ovisits)
    orc.orc_lens_destroy(lens)
    ctx = gpu_ctx_factory()
    for _ in range(2):          # (twice: the second pass of a context runs blind / streamed, and "no-log" keeps its log by then)
        c = gpu_run(ctx, p, table, visits, n_aovs=3, kinds=kinds, log_cap=(1 << 22) if own_log else 0)
        _same_counters(c, ref)
        if own_log:
            check_logs(ctx, ref)
        check_frame(ctx, ref, n_aovs=3, kinds=kinds)
        for a in (1, 2):        # the closest-filtered planes: the oracle's values, bit for bit
            assert np.array_equal(ctx.download_aov(a).view(np.uint32), ref.resolve(a).view(np.uint32))
    ref.close()
    # gaussian AOVs only: the depth of a sample is never looked at
    ref = common.run_oracle(orc, p, table, ovisits, n_aovs=3)
    ctx2 = gpu_ctx_factory()
    c = gpu_run(ctx2, p, table, ovisits, n_aovs=3, kinds=[0, 0, 0])
    _same_counters(c, ref)
    check_logs(ctx2, ref)
    check_frame(ctx2, ref, n_aovs=3)
