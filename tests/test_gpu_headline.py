"""The pass bench.py times, at the size it is timed at, against the oracle.

bench.py's headline step is the *streamed* pass -- scan_dma_kernel publishing to persistent solve waves, live
stragglers, the second round beside the first accept, the early resolve -- on 3840x2160 / 9 visits per pixel /
1024 draws / f_hi 2^-16, alternating between two seeded streams.  Here that very workload (same generator on the
device, same seeds, same order of passes) is compared with the oracle run over the whole frame, threaded over
rows: accepted-draw lists bit-identical, counters equal, direct sums bit-exact where no draw lands, everything
within 1e-5 where draws land, including the image the pass resolves on its way.
Reference loop being replaced: src/lentil_filter.cpp:248-299; direct adds src/lentil.h:938-955.
"""
import numpy as np
import pytest

import common
from pota_amd import capi, workload
from test_gpu_parity import check_frame, check_logs

pytestmark = pytest.mark.gpu


def _device_stream(torch, p, W, H, M, seed, f_hi, v_begin, v_end, y0=0, n_extra=0):
    """The bench's generator on the device + the same arrays on the host (the oracle reads exactly what the GPU reads)."""
    dev = torch.device("cuda:0")
    cols = workload.generate(torch, v_begin, v_end, W, H, M, seed=seed, f_hi=f_hi, focus_dist=150.0,
                             tan_half_fov=common.tan_half_fov(p), device=dev, n_extra=n_extra)
    dv, dkeep = capi.make_visits(cols, visits_per_pixel=M, pixels_per_row=W, pixel_y0=y0, ptr=lambda t: t.data_ptr())
    torch.cuda.synchronize()
    host = {k: (v.cpu().numpy() if k != "extra" else [e.cpu().numpy() for e in v]) for k, v in cols.items()}
    hv, hkeep = capi.make_visits(host, visits_per_pixel=M, pixels_per_row=W, pixel_y0=y0)
    return (dv, (cols, dkeep)), (hv, (host, hkeep))


def _oracle_threads(p, per_frame_gb):
    """(common.ThreadedOracle keeps ONE frame whatever the thread count -- orc_redistribute_threads; per_frame_gb only bounds
    the fallback with a private frame per thread, which none of this file's streams takes)"""
    import os
    return int(max(2, min(64, os.cpu_count() or 2)))


def _compare(ctx, ref, c, samples, n_visits, p, n_aovs=1, kinds=None):
    rc = ref.counters()
    assert (c.visits, c.redistributed_visits, c.attempted_draws, c.accepted_draws) == (
        n_visits, rc.redistributed_visits, rc.attempted_draws, rc.accepted_draws)
    assert c.worklist_overflow == 0
    check_logs(ctx, ref)                      # (visit, attempt, pixel) of every accepted draw, bit for bit
    st = {}
    worst = check_frame(ctx, ref, n_aovs=n_aovs, kinds=kinds, stats=st)      # accumulators, weights, resolved image (the early-resolved one): 1e-5
    # ... and BASELINE.json's bar to the letter for the configurations it names: every accumulator and every resolved value within
    # 1e-5 (relative) of the reference CPU imager's own fp32 number -- not of the exact sum plus what the fp32 number is off
    # (check_frame's assertion, which a frame whose draws pile up by the thousand on one pixel needs; none of these does:
    # 5.0e-6 headline, 5.8e-6 config 3, 2.6e-6 config 4, 1.1e-6 config 2 in bench.py's in-run checks)
    assert st["vs_fp32"] <= 1e-5, st
    # pixels no draw lands on only hold their own visits, added in iterator order: bit-exact
    touched = np.zeros(p.xres * p.yres, bool)
    touched[ref.log()[:, 2]] = True
    rec = ctx.download_records()              # every AOV's accumulators and the weight, one copy
    buf, w = rec[:, 0:4], rec[:, 4 * n_aovs]
    assert np.array_equal(buf[~touched], ref.buffer(0)[~touched])
    assert np.array_equal(w[~touched], ref.weight()[~touched])
    assert np.array_equal(ctx.download_aov(0)[~touched], ref.resolve(0)[~touched])
    for a in range(1, n_aovs):
        if kinds is not None and kinds[a] != 0:
            # a closest-filtered AOV is a copy of one candidate's value (src/lentil.h:832-837): the whole frame bit for bit
            assert np.array_equal(ctx.download_aov(a), ref.resolve(a)), a
        else:
            assert np.array_equal(rec[:, 4 * a:4 * a + 4][~touched], ref.buffer(a)[~touched]), a
    return worst, int(touched.sum())


def _timed_config_vs_oracle(orc, name, W, H, lens, S, n_extra=0, kinds=None, bokeh_image=False, seeds=(0x5EED, 0xBEEF),
                            passes=(0, 1, 0, 1), runtime_kernel=False):
    """One of bench.py's `configs` entries as the bench runs it -- Bench.generate's two seeded streams, two set-up passes,
    then alternating streams -- with every pass compared with the oracle over the whole frame."""
    import ctypes as C
    import os
    import torch
    from pota_amd import _abi, bokeh
    M, f_hi = 9, 2.0 ** -16
    n = W * H * M
    n_aovs = 1 + n_extra
    has_closest = bool(kinds) and any(k != 0 for k in kinds)
    per_thread_gb = W * H * n_aovs * 60e-9 + 0.2
    if common.host_memory_gb() < 2 * per_thread_gb + 30:
        pytest.skip("not enough host memory for the oracle's frames")
    kw = dict(bokeh_enable_image=1) if bokeh_image else {}
    p, model, table, keep = common.po_setup(W, H, lens=lens, samples_override=S, **kw)
    tables = ob = None
    if bokeh_image:
        tex = np.load(os.path.join(common.ROOT, "tests", "golden", "example_bokeh_kernel_u8.npy")).astype(np.float32) / np.float32(255)
        tables = bokeh.build_tables(tex)
        bt = _abi.BokehTable()
        bt.x, bt.y = tables["x"], tables["y"]
        for k in ("cdfRow", "rowIndices", "cdfColumn", "columnIndices"):
            setattr(bt, k, tables[k].ctypes.data)
        ob = orc.orc_bokeh_from_tables(C.byref(bt))
    ctx = capi.Context(0)
    try:
        ctx.set_params(p); ctx.set_lens(table); ctx.set_bokeh(tables)
        if runtime_kernel:
            # a table with no kernel built into the library: its solve kernels are emitted and compiled at run time (lentil_lens_jit.h;
            # seconds from the cache on disk, ~15 s the first time on a box) -- the passes below must run THEM, not the interpreter
            assert not ctx.lens_is_compiled()
            ctx.lens_jit_wait(600.0)
            assert ctx.lens_jit_status()[0] == 2, ctx.lens_jit_status()
        else:
            assert ctx.lens_is_compiled()
        ctx.alloc_frame(n_aovs, kinds)
        ctx.set_draw_log(1 << 21)
        streams, refs = [], []
        for seed in seeds:
            d, h = _device_stream(torch, p, W, H, M, seed, f_hi, 0, n, n_extra=n_extra)
            streams.append(d)
            refs.append(common.ThreadedOracle(orc, p, table, h[0], _oracle_threads(p, per_thread_gb), n_aovs=n_aovs,
                                              kinds=kinds, bokeh=ob))
            del h
        seen_streamed = 0
        for k, i in enumerate(passes):
            dv, dkeep = streams[i % len(streams)]
            ctx.bind_visits(dv, dkeep)
            ctx.clear_frame(); ctx.redistribute(); ctx.resolve(); ctx.sync()
            c = ctx.counters()
            # (the first pass of a context looks at its scans; with extra AOVs the streamed pass is scan_dma_multi_kernel's,
            # which takes gaussian AOVs only -- closest-filtered ones keep the chunked form, blind from the second pass on)
            want = 1 if (k and not has_closest) else 0
            assert c.fallback_chunks == 0 and c.streamed == want, (k, c.streamed, c.fallback_chunks, ctx.last_redo_note())
            seen_streamed += c.streamed
            worst, n_touched = _compare(ctx, refs[i % len(refs)], c, S, n, p, n_aovs=n_aovs, kinds=kinds)
            print("%s, pass %d (%s): %d items, %d accepted draws on %d pixels, max rel err %.2e"
                  % (name, k, "streamed" if c.streamed else "chunked", c.redistributed_visits, c.accepted_draws, n_touched, worst))
        assert seen_streamed == (0 if has_closest else len(passes) - 1)
    finally:
        ctx.close()
        for r in refs:
            r.close()
        if ob:
            orc.orc_bokeh_destroy(ob)


def test_config2_1080p_256_draws_vs_oracle(orc):
    """BASELINE config 2 (double-gauss 50mm, 1920x1080, 256 draws, beauty only) as bench.py's `configs` entry runs it."""
    _timed_config_vs_oracle(orc, "config 2", 1920, 1080, "double_gauss_50mm", 256)


def test_config3_4k_512_draws_aperture_image_vs_oracle(orc):
    """BASELINE config 3 (double-gauss 50mm + the aperture image's CDF draws, src/imagebokeh.h:341-412; 3840x2160, 512
    draws) as bench.py's `configs` entry runs it."""
    _timed_config_vs_oracle(orc, "config 3", 3840, 2160, "double_gauss_50mm", 512, bokeh_image=True)


def test_config4_4k_petzval_two_closest_aovs_vs_oracle(orc):
    """BASELINE config 4 (petzval, 3840x2160, 1024 draws, beauty + 8 AOVs) at full size with two of the eight
    closest-filtered (Camera::add_to_buffer's z-test, src/lentil.h:832-837): the register-staged scan, the key plane, the
    gather -- every pass against the oracle, the closest AOVs bit for bit over the whole frame.  One stream (the oracle's
    nine-AOV frames are what bounds this test), replayed as the bench's passes are."""
    _timed_config_vs_oracle(orc, "config 4 (two closest AOVs)", 3840, 2160, "petzval_58mm", 1024, n_extra=8,
                            kinds=[0, 0, 1, 0, 0, 0, 1, 0, 0], seeds=(0x5EED,), passes=(0, 0))      # (first pass, then the blind form it keeps)


def test_config4_anamorphic_4k_nine_aovs_runtime_kernel_vs_oracle(orc):
    """BASELINE config 4's other half, "anamorphic": anamorphic_petzval_58mm -- a cylindrical outer pupil (src/lens.h:156-221,
    cylinderToCs / csToCylinder), no kernel of it built into the library -- at the full 3840x2160 / 1024 draws / beauty + 8
    gaussian AOVs, through the solve kernels the library compiles for the table at run time: the whole frame against the
    oracle, the context's first pass (chunked) and a streamed one."""
    _timed_config_vs_oracle(orc, "config 4 (anamorphic, run-time kernel)", 3840, 2160, "anamorphic_petzval_58mm", 1024, n_extra=8,
                            seeds=(0x5EED,), passes=(0, 0), runtime_kernel=True)


def test_headline_4k_streamed_vs_oracle(orc):
    import torch
    W, H, M, S, f_hi = 3840, 2160, 9, 1024, 2.0 ** -16
    n = W * H * M
    p, model, table, keep = common.po_setup(W, H, samples_override=S)
    ctx = capi.Context(0)
    try:
        ctx.set_params(p); ctx.set_lens(table); ctx.set_bokeh(None)
        assert ctx.lens_is_compiled()
        ctx.alloc_frame(1)
        ctx.set_draw_log(1 << 21)
        streams, refs = [], []
        threads = _oracle_threads(p, 0.7)
        for seed in (0x5EED, 0xBEEF):               # bench.py's SEEDS
            d, h = _device_stream(torch, p, W, H, M, seed, f_hi, 0, n)
            streams.append(d)
            ref = common.ThreadedOracle(orc, p, table, h[0], threads)
            assert 900 < ref.counters().redistributed_visits < 1400
            refs.append(ref)
            del h
        # bench.py's order: set-up passes, then alternating streams.  The first pass of a context looks at its scans
        # (chunked); every later one is streamed.
        for i, want_streamed in ((0, 0), (1, 1), (0, 1), (1, 1)):
            dv, dkeep = streams[i]
            ctx.bind_visits(dv, dkeep)
            ctx.clear_frame(); ctx.redistribute(); ctx.resolve(); ctx.sync()
            c = ctx.counters()
            assert c.streamed == want_streamed and c.fallback_chunks == 0, (i, c.streamed, c.fallback_chunks, ctx.last_redo_note())
            worst, n_touched = _compare(ctx, refs[i], c, S, n, p)
            print("stream %d %s: %d items, %d accepted draws on %d pixels, max rel err %.2e"
                  % (i, "streamed" if want_streamed else "chunked", c.redistributed_visits, c.accepted_draws, n_touched, worst))
    finally:
        ctx.close()


@pytest.mark.parametrize("form", ["chunked", "streamed"])
def test_config5_quarter_frame_vs_oracle(orc, monkeypatch, form):
    """BASELINE config 5 (7680x4320, 2048 draws) on a quarter of the frame, the 1080 rows around the optical axis, at the
    full frame's geometry -- in the chunked form (blind from the second pass on) and in the streamed one, which is what one
    GPU runs the whole frame in since round 3 (fewer than one draw per 24 visits)."""
    import torch
    W, H, M, S, f_hi = 7680, 4320, 9, 2048, 2.0 ** -16
    y0, rows = 1620, 1080
    p, model, table, keep = common.po_setup(W, H, samples_override=S)
    monkeypatch.setenv("LENTIL_STREAM", "0" if form == "chunked" else "1")
    ctx = capi.Context(0)
    try:
        ctx.set_params(p); ctx.set_lens(table); ctx.set_bokeh(None)
        ctx.alloc_frame(1)
        ctx.set_draw_log(1 << 22)
        d, h = _device_stream(torch, p, W, H, M, 0x5EED, f_hi, y0 * W * M, (y0 + rows) * W * M, y0=y0)
        ref = common.ThreadedOracle(orc, p, table, h[0], _oracle_threads(p, 2.6))
        del h
        n = rows * W * M
        for blind in (0, 2):
            ctx.bind_visits(*d)
            ctx.clear_frame(); ctx.redistribute(); ctx.resolve(); ctx.sync()
            c = ctx.counters()
            if form == "chunked":
                assert c.streamed == 0 and c.blind_chunks == blind and c.fallback_chunks == 0
            else:
                assert c.streamed == (1 if blind else 0) and c.fallback_chunks == 0
            worst, n_touched = _compare(ctx, ref, c, S, n, p)
            print("config 5 quarter frame, %s: %d items, %d accepted draws on %d pixels, max rel err %.2e"
                  % ("blind" if blind else "first pass", c.redistributed_visits, c.accepted_draws, n_touched, worst))
    finally:
        ctx.close()


def test_config4_like_4k_nine_gaussian_aovs_streamed_vs_oracle(orc):
    """BASELINE config 4's geometry -- petzval table, 3840x2160, 1024 draws, beauty + 8 AOVs -- with every AOV gaussian,
    which is what takes scan_dma_multi_kernel and, from the second pass of a context on, the streamed pass: the whole
    frame against the oracle, first pass (chunked) and two streamed ones.  (With closest-filtered AOVs among the eight the
    register-staged scan runs: test_config4_4k_petzval_8_aovs.)"""
    import torch
    W, H, M, S, f_hi, K = 3840, 2160, 9, 1024, 2.0 ** -16, 8
    n = W * H * M
    per_thread_gb = W * H * (K + 1) * 60e-9 + 0.2
    if common.host_memory_gb() < 2 * per_thread_gb + 40:
        pytest.skip("not enough host memory for the oracle's nine-AOV frames")
    p, model, table, keep = common.po_setup(W, H, lens="petzval_58mm", samples_override=S)
    ctx = capi.Context(0)
    try:
        ctx.set_params(p); ctx.set_lens(table); ctx.set_bokeh(None)
        assert ctx.lens_is_compiled()
        ctx.alloc_frame(K + 1)
        ctx.set_draw_log(1 << 21)
        d, h = _device_stream(torch, p, W, H, M, 0x5EED, f_hi, 0, n, n_extra=K)
        ref = common.ThreadedOracle(orc, p, table, h[0], _oracle_threads(p, per_thread_gb), n_aovs=K + 1)
        del h
        assert 900 < ref.counters().redistributed_visits < 1400
        for want_streamed in (0, 1, 1):
            ctx.bind_visits(*d)
            ctx.clear_frame(); ctx.redistribute(); ctx.resolve(); ctx.sync()
            c = ctx.counters()
            assert c.streamed == want_streamed and c.fallback_chunks == 0, (c.streamed, c.fallback_chunks, ctx.last_redo_note())
            worst, n_touched = _compare(ctx, ref, c, S, n, p, n_aovs=K + 1)
            print("nine AOVs %s: %d items, %d accepted draws on %d pixels, max rel err %.2e"
                  % ("streamed" if want_streamed else "chunked", c.redistributed_visits, c.accepted_draws, n_touched, worst))
    finally:
        ctx.close()
