"""The asynchronous end of a pass (include/lentil_hip.h, lentil_hip_set_async / lentil_hip_pass_totals; round 6).

lentil_hip_redistribute returns once a streamed pass's kernels are enqueued; the verdict about the pass -- did it fit, does it
need another round, did a wave give up waiting -- is read by the next call that observes the context.  What these tests hold
it to: (1) observed, a pass is exactly what it was when redistribute still waited (draw lists bit for bit, 1e-5 radiance,
against the oracle), including when it needs its draws redone and when other visits have been bound in between;
(2) pipelined (clear, pass, resolve, clear ... with no observation in between) every pass still does all of its work, the
totals add up to the oracle's, and the frame observed at the end is right; (3) a pass cleared away while it still needed
work is counted, and the passes behind it are unharmed.

The reference has no counterpart (its imager is one synchronous loop, src/lentil_imager.cpp:66-193); the oracle the frames are
compared with is oracle/lentil_oracle.cpp as everywhere else.
"""
import numpy as np
import pytest

import common
from pota_amd import capi
from test_gpu_parity import check_frame, check_logs

pytestmark = pytest.mark.gpu


def _device_stream(torch, cols, M, W):
    """the columns of a host stream in device memory, bound by pointer (nothing is copied or freed by the library)"""
    dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in cols.items() if k != "extra" and v is not None}
    dev["extra"] = [torch.from_numpy(np.ascontiguousarray(e)).cuda() for e in cols.get("extra", [])]
    v, keep = capi.make_visits(dev, visits_per_pixel=M, pixels_per_row=W, ptr=lambda t: t.data_ptr())
    torch.cuda.synchronize()
    return v, dev


def _setup(orc, W, H, M, S, f_a, f_b):
    import torch
    p, model, table, keep = common.po_setup(W, H, samples_override=S)
    out = []
    for f_hi, seed in ((f_a, 0x5EED), (f_b, 0xBEEF)):
        hv, cols = common.make_stream(p, W, H, M, f_hi=f_hi, seed=seed)
        ref = common.ThreadedOracle(orc, p, table, hv, 8)
        dv, dkeep = _device_stream(torch, cols, M, W)
        out.append((dv, dkeep, ref, hv, cols))
    ctx = capi.Context(0)
    ctx.set_params(p)
    ctx.set_lens(table)
    ctx.alloc_frame(1)
    ctx.set_draw_log(1 << 22)
    return ctx, p, table, keep, out


def _frame(ctx, v):
    ctx.bind_visits(v)
    ctx.clear_frame()
    ctx.redistribute()
    ctx.resolve()


@pytest.mark.parametrize("asynchronous", [True, False])
def test_pipelined_frames_do_all_their_work(orc, asynchronous):
    """Eight frames alternating two visit streams, nothing observed in between: the library's totals must be the oracle's
    sums, the passes must have run deferred (asynchronous) or not at all so (set_async(0)), and the last frame -- observed --
    is the oracle's, draw for draw."""
    W, H, M, S = 640, 360, 9, 128
    ctx, p, table, keep, streams = _setup(orc, W, H, M, S, 2.0 ** -12, 2.0 ** -12)
    try:
        ctx.set_async(asynchronous)
        for k in range(3):                         # the context's first passes size its buffers (not streamed, then streamed)
            _frame(ctx, streams[k & 1][0])
        ctx.sync()
        ctx.pass_totals(reset=True)
        K = 8
        for k in range(K):
            _frame(ctx, streams[k & 1][0])
        t = ctx.pass_totals(reset=True)            # observes: waits, looks at the last pass's end
        assert int(t.passes) == K and int(t.streamed) == K and int(t.fallback_chunks) == 0 and int(t.worklist_overflow) == 0
        if asynchronous:
            assert int(t.deferred) == K and int(t.abandoned) == K - 1 and int(t.abandoned_incomplete) == 0
        else:
            assert int(t.deferred) == 0 and int(t.abandoned) == 0
        want = [streams[k & 1][2].counters() for k in range(K)]
        assert int(t.redistributed_visits) == sum(int(c.redistributed_visits) for c in want)
        assert int(t.attempted_draws) == sum(int(c.attempted_draws) for c in want)
        assert int(t.accepted_draws) == sum(int(c.accepted_draws) for c in want)
        assert t.scan_ms > 0 and t.draw_ms > 0
        ref = streams[(K - 1) & 1][2]
        check_logs(ctx, ref)
        check_frame(ctx, ref)
    finally:
        ctx.close()
        for s in streams:
            s[2].close()


@pytest.mark.parametrize("observed", [True, False])
def test_a_deferred_pass_that_needs_its_draws_redone(orc, observed):
    """light, light, then a stream with sixty times the highlights: the pass sized from the light one does not fit and returns
    before anybody knows.  Observed -- after OTHER visits have been bound, which is what a pipelining caller does first -- it is
    redone with exact sizes on the visits it ran on and gives the oracle's frame.  Cleared away instead, it is counted as
    abandoned incomplete, and the same heavy stream run behind it -- sized from the abandoned pass's counters -- is whole."""
    W, H, M, S = 96, 64, 9, 48
    ctx, p, table, keep, streams = _setup(orc, W, H, M, S, 0.002, 0.12)
    (light, _, ref_l, _, _), (heavy, _, ref_h, _, _) = streams
    try:
        for _ in range(2):
            _frame(ctx, light)
        ctx.sync()
        ctx.pass_totals(reset=True)
        if observed:
            _frame(ctx, heavy)
            ctx.bind_visits(light)                     # the caller moves on ...
            c = ctx.counters()                         # ... and only then observes: the heavy pass's end
            rc = ref_h.counters()
            assert (c.redistributed_visits, c.attempted_draws, c.accepted_draws) == (rc.redistributed_visits, rc.attempted_draws, rc.accepted_draws)
            assert c.fallback_chunks >= 1 and c.worklist_overflow == 0
            check_logs(ctx, ref_h)
            check_frame(ctx, ref_h)
            t = ctx.pass_totals(reset=True)
            assert int(t.passes) == 1 and int(t.deferred) == 1 and int(t.abandoned) == 0 and int(t.fallback_chunks) >= 1
        else:
            _frame(ctx, heavy)                         # does not fit, nobody looks
            _frame(ctx, heavy)                         # its clear abandons the first; sized from the light passes or from the first one's counters
            ctx.sync()                                 # (observes the second: whatever it needed is done)
            _frame(ctx, heavy)                         # sized from what the heavy passes found: fits
            t = ctx.pass_totals(reset=True)
            assert int(t.passes) == 3 and int(t.abandoned) == 1 and int(t.abandoned_incomplete) == 1, (
                int(t.passes), int(t.abandoned), int(t.abandoned_incomplete))
            assert int(t.worklist_overflow) == 0
            check_logs(ctx, ref_h)
            check_frame(ctx, ref_h)
            c = ctx.counters()
            assert c.fallback_chunks == 0 and c.streamed == 1
    finally:
        ctx.close()
        ref_l.close()
        ref_h.close()


def test_two_passes_into_one_frame_and_the_resolve_behind_a_deferred_pass(orc):
    """redistribute twice without a clear: the second pass observes the first (it adds to the same frame), and the frame
    holds both; a resolve enqueued behind a deferred pass shows the pass's whole result once observed."""
    W, H, M, S = 96, 64, 9, 48
    ctx, p, table, keep, streams = _setup(orc, W, H, M, S, 0.004, 0.004)
    (a, _, ref_a, _, _), (b, _, ref_b, _, _) = streams
    try:
        for _ in range(2):
            _frame(ctx, a)
        _frame(ctx, a)
        img1 = ctx.download_aov(0)                 # observes
        assert np.allclose(img1, ref_a.resolve(0), rtol=1e-5, atol=0)
        ctx.bind_visits(b)
        ctx.redistribute()                         # second pass into the frame that holds a's
        ctx.resolve()
        buf, w = ctx.download_accum(0)
        want = ref_a.buffer64(0) + ref_b.buffer64(0)
        m = want != 0
        assert np.array_equal(buf != 0, m)
        assert float(np.max(np.abs(buf[m] - want[m]) / np.abs(want[m]))) < 2e-5
    finally:
        ctx.close()
        ref_a.close()
        ref_b.close()
