"""Occlusion probes (include/lentil_hip.h: lentil_hip_set_occlusion_probe; round 6).

The reference asks the renderer before every backward trace whether anything stands between the sample and the point of the
aperture the trace goes through (AiTraceProbe; src/lentil.h:613-629 for polynomial optics, src/lentil_filter.cpp:356-375 for
the thin lens), and an occluded try fails.  The HIP path has no scene: once a round's traces are solved it hands the host the
segments of the tries that got through the lens, the host answers a byte each, the accept honours it.  Here the "scene" is an
analytic sphere (oracle/lentil_oracle.cpp: orc_sphere_occluder -- a plain C lentil_probe_fn), given to the HIP library and to
the oracle alike: accepted-draw lists bit for bit, radiance at 1e-5, and the probe must actually bite (draws differ from the
unoccluded frame's).
"""
import ctypes as C

import numpy as np
import pytest

import common
import oracle_lib
from pota_amd import _abi, capi
from test_gpu_parity import check_frame, check_logs

pytestmark = pytest.mark.gpu


def _run_gpu(ctx, p, table, visits, probe, n_aovs=1, passes=2):
    ctx.set_params(p)
    if table is not None:
        ctx.set_lens(table)
    ctx.set_bokeh(None)
    ctx.alloc_frame(n_aovs)
    ctx.set_draw_log(1 << 22)
    ctx.set_occlusion_probe(*probe)
    ctx.upload_visits(visits)
    for _ in range(passes):                     # (the second pass of a context would run streamed: with a probe it may not)
        ctx.clear_frame(); ctx.redistribute(); ctx.resolve(); ctx.sync()
        c = ctx.counters()
        assert c.worklist_overflow == 0 and c.streamed == 0
    return c


@pytest.mark.parametrize("camera", ["po", "thinlens"])
def test_occluder_between_the_samples_and_the_lens(orc, camera):
    W, H, M, S = 96, 64, 9, 48
    if camera == "po":
        p, model, table, keep = common.po_setup(W, H, samples_override=S)
    else:
        p, table = common.tl_setup(W, H, samples_override=S), None
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.02, n_extra=1)
    # a sphere beside the optical axis, between the lens and the far highlights (camera at the origin looking down -z, cm)
    sphere = np.array([6.0, 2.0, -70.0, 9.0], np.float32)
    probe = (oracle_lib.sphere_occluder(orc), sphere.ctypes.data)
    ref = common.ThreadedOracle(orc, p, table, visits, 4, n_aovs=2, probe=probe)
    free = common.ThreadedOracle(orc, p, table, visits, 4, n_aovs=2)
    ctx = capi.Context(0)
    try:
        c = _run_gpu(ctx, p, table, visits, probe, n_aovs=2)
        rc = ref.counters()
        assert (c.redistributed_visits, c.attempted_draws, c.accepted_draws) == (rc.redistributed_visits, rc.attempted_draws, rc.accepted_draws)
        check_logs(ctx, ref)
        check_frame(ctx, ref, n_aovs=2)
        probed, occluded, calls = ctx.probe_stats()
        assert probed > 1000 and 0 < occluded < probed and calls >= 2, (probed, occluded, calls)
        # ... and the occluder changed something: draws the unoccluded frame has are gone or went elsewhere
        fl, rl = common.sort_log(free.log()), common.sort_log(ref.log())
        assert fl.shape != rl.shape or not np.array_equal(fl, rl)
        assert free.counters().attempted_draws != rc.attempted_draws or free.counters().accepted_draws != rc.accepted_draws or fl.shape == rl.shape
        # probing off again: the unoccluded frame, and the pass may stream again
        ctx.set_occlusion_probe(None)
        for _ in range(2):
            ctx.clear_frame(); ctx.redistribute(); ctx.resolve(); ctx.sync()
        check_logs(ctx, free)
        check_frame(ctx, free, n_aovs=2)
    finally:
        ctx.close()
        ref.close()
        free.close()


def test_probe_with_a_given_camera_to_world_and_a_moved_camera(orc):
    """A camera that is not at the origin: world_to_camera a translation + rotation about y; the probe's segments are world-space
    ones (the sphere sits in the world), with AiCameraToWorldMatrix given explicitly on one side and computed (fp64 inverse of
    world_to_camera) on the other -- the same frame either way, and the oracle's."""
    W, H, M, S = 96, 64, 9, 32
    p, model, table, keep = common.po_setup(W, H, samples_override=S)
    a = np.float32(0.1)
    rot = np.array([[np.cos(a), 0, -np.sin(a), 0], [0, 1, 0, 0], [np.sin(a), 0, np.cos(a), 0], [0, 0, 0, 1]], np.float64)
    tr = np.eye(4); tr[3, :3] = (5.0, -3.0, 20.0)
    c2w = (rot @ tr).astype(np.float32)                       # row-vector convention: p_world = p_cam @ c2w
    w2c = np.linalg.inv(c2w.astype(np.float64)).astype(np.float32)
    for r in range(4):
        for c in range(4):
            p.world_to_camera[r][c] = float(w2c[r, c])
    # samples generated in camera space, moved into the world
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.02)
    pos = cols["pos_z"]
    ph = np.concatenate([pos[:, :3].astype(np.float64), np.ones((pos.shape[0], 1))], axis=1) @ c2w.astype(np.float64)
    pos[:, :3] = ph[:, :3].astype(np.float32)
    sphere_cam = np.array([6.0, 2.0, -70.0, 1.0])
    sw = sphere_cam @ c2w.astype(np.float64)
    sphere = np.array([sw[0], sw[1], sw[2], 9.0], np.float32)
    fn = oracle_lib.sphere_occluder(orc)
    ref = common.ThreadedOracle(orc, p, table, visits, 4, probe=(fn, sphere.ctypes.data, c2w))
    ctx = capi.Context(0)
    try:
        _run_gpu(ctx, p, table, visits, (fn, sphere.ctypes.data, c2w), passes=1)
        check_logs(ctx, ref)
        check_frame(ctx, ref)
        assert ctx.probe_stats()[1] > 0
        # the inverse computed by the library: the same matrix up to rounding -- the frame must still be the oracle's computed the same way
        ref2 = common.ThreadedOracle(orc, p, table, visits, 4, probe=(fn, sphere.ctypes.data))
        _run_gpu(ctx, p, table, visits, (fn, sphere.ctypes.data), passes=1)
        check_logs(ctx, ref2)
        check_frame(ctx, ref2)
        ref2.close()
    finally:
        ctx.close()
        ref.close()


def test_probe_is_refused_with_chromatic_aberration(orc):
    W, H, M = 32, 24, 9
    p, model, table, keep = common.po_setup(W, H, samples_override=16, abb_chromatic=0.5)
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.05)
    sphere = np.array([0, 0, -70.0, 5.0], np.float32)
    ctx = capi.Context(0)
    try:
        ctx.set_params(p); ctx.set_lens(table); ctx.alloc_frame(1)
        ctx.set_occlusion_probe(oracle_lib.sphere_occluder(orc), sphere.ctypes.data)
        ctx.upload_visits(visits)
        ctx.clear_frame()
        with pytest.raises(capi.LentilError) as e:
            ctx.redistribute()
        assert e.value.code == _abi.ERR_UNSUPPORTED
    finally:
        ctx.close()
