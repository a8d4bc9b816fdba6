"""The visit stream handed over piece by piece (lentil_hip_visits_begin / _append / _wait / _end, lentil_hip_host_alloc):
the pass over a stream assembled from blocks must be the pass over the same stream uploaded in one piece
(lentil_hip_upload_visits), which the parity suite pins to the oracle."""
import ctypes as C
import threading

import numpy as np
import pytest

import common
from pota_amd import _abi, capi
from test_gpu_parity import TOL, check_frame, check_logs, gpu_run

pytestmark = pytest.mark.gpu
COLS = ("rgba", "pos_z", "raydir_time", "volume_ignore", "transmission")


class _PinnedBlock:
    """one page-locked block holding `cap` visits of every column"""

    def __init__(self, cap, n_extra, ragged):
        self.cap, self.n_extra, self.ragged = cap, n_extra, ragged
        n_col = 5 + n_extra
        self.bytes = cap * (16 * n_col + (8 if ragged else 0))
        self.ptr = capi.host_alloc(self.bytes)
        buf = (C.c_char * self.bytes).from_address(self.ptr)
        self.f = np.frombuffer(buf, np.float32, cap * 4 * n_col).reshape(n_col, cap, 4)
        if ragged:
            off = cap * 16 * n_col
            self.pixel = np.frombuffer(buf, np.uint32, cap, off)
            self.inv = np.frombuffer(buf, np.float32, cap, off + 4 * cap)

    def fill(self, cols, lo, hi):
        n = hi - lo
        for c, name in enumerate(COLS):
            self.f[c, :n] = cols[name][lo:hi]
        for k in range(self.n_extra):
            self.f[5 + k, :n] = cols["extra"][k][lo:hi]
        v = _abi.Visits()
        v.n, v.n_extra = n, self.n_extra
        for c, name in enumerate(COLS):
            setattr(v, name, self.f[c].ctypes.data)
        for k in range(self.n_extra):
            v.extra[k] = self.f[5 + k].ctypes.data
        if self.ragged:
            self.pixel[:n] = cols["pixel"][lo:hi]
            self.inv[:n] = cols["inv_density"][lo:hi]
            v.pixel, v.inv_density = self.pixel.ctypes.data, self.inv.ctypes.data
        return v

    def free(self):
        self.f = self.pixel = self.inv = None
        capi.host_free(self.ptr)


def _layout(visits):
    lay = _abi.Visits()
    for name in ("visits_per_pixel", "pixels_per_row", "pixel_x0", "pixel_y0", "pixel_row_stride", "n_extra"):
        setattr(lay, name, getattr(visits, name))
    lay.inv_density = visits.inv_density          # non-NULL: per-visit densities
    return lay


def _prepare(ctx, p, table, n_aovs):
    ctx.set_params(p)
    if table is not None:
        ctx.set_lens(table)
    ctx.set_bokeh(None)
    ctx.alloc_frame(n_aovs)
    ctx.set_draw_log(1 << 22)


def _pass(ctx):
    ctx.clear_frame(); ctx.redistribute(); ctx.resolve(); ctx.sync()
    return ctx.counters()


def test_uniform_stream_in_blocks_equals_the_whole_upload(orc, gpu_ctx_factory):
    """Uniform footprint, two AOVs, blocks of odd sizes from two pinned blocks used in turn (a block is refilled after
    its ticket has been waited for), a capacity hint far too small (the columns grow twice), then a second frame that
    reuses the columns."""
    W, H, M = 64, 48, 9
    p, model, table, keep = common.po_setup(W, H, samples_override=48)
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.02, n_extra=1)
    ref = common.run_oracle(orc, p, table, visits, n_aovs=2)
    n = int(visits.n)
    ctx = gpu_ctx_factory()
    _prepare(ctx, p, table, 2)
    blocks = [_PinnedBlock(5000, 1, False) for _ in range(2)]
    try:
        for frame in range(2):
            ctx.visits_begin(_layout(visits), capacity_hint=n // 5 if frame == 0 else 0)
            tickets = [0, 0]
            lo, k = 0, 0
            rng = np.random.default_rng(frame)
            while lo < n:
                hi = min(n, lo + int(rng.integers(1, 5001)))
                b = k % 2
                ctx.visits_wait(tickets[b])
                tickets[b] = ctx.visits_append(blocks[b].fill(cols, lo, hi))
                lo, k = hi, k + 1
            assert ctx.visits_end() == n
            c = _pass(ctx)
            rc = ref.counters()
            assert (c.redistributed_visits, c.attempted_draws, c.accepted_draws) == (
                rc.redistributed_visits, rc.attempted_draws, rc.accepted_draws)
            check_logs(ctx, ref)
            check_frame(ctx, ref, n_aovs=2)
    finally:
        for b in blocks:
            b.free()


def test_ragged_stream_appended_from_threads(orc, gpu_ctx_factory):
    """Ragged footprint with per-visit densities, appended by four threads in whatever order they get to it (as bucket
    threads would): same counters, accumulators within the fp32 summation-order tolerance of the one-piece upload of
    the same visits -- and ordinary (pageable) memory as the source."""
    W, H, M = 64, 48, 9
    p, model, table, keep = common.po_setup(W, H, samples_override=32)
    uvisits, cols = common.make_stream(p, W, H, M, f_hi=0.02)
    n = int(uvisits.n)
    idx = np.arange(n, dtype=np.uint32) // M
    cols = dict(cols)
    cols["pixel"] = ((idx % W) | ((idx // W) << 16)).astype(np.uint32)
    cols["inv_density"] = np.random.default_rng(3).choice(np.array([1.0 / 9, 1.0 / 4, 1.0], np.float32), n).astype(np.float32)
    visits, keepv = capi.make_visits(cols, visits_per_pixel=0)
    whole = gpu_ctx_factory()
    _prepare(whole, p, table, 1)
    whole.upload_visits(visits)
    cw = _pass(whole)
    assert cw.redistributed_visits > 100

    ctx = gpu_ctx_factory()
    _prepare(ctx, p, table, 1)
    ctx.visits_begin(_layout(visits), capacity_hint=n)
    cuts = np.linspace(0, n, 4 * 6 + 1).astype(int)
    errors = []

    def worker(t):
        try:
            for j in range(t, len(cuts) - 1, 4):
                lo, hi = int(cuts[j]), int(cuts[j + 1])
                part, kp = capi.make_visits({k: (v[lo:hi] if k != "extra" else []) for k, v in cols.items()}, visits_per_pixel=0)
                ctx.visits_wait(ctx.visits_append(part))        # pageable source: keep it alive until copied
        except Exception as e:          # pragma: no cover
            errors.append(e)

    th = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not errors, errors
    assert ctx.visits_end() == n
    c = _pass(ctx)
    assert (c.redistributed_visits, c.attempted_draws, c.accepted_draws) == (
        cw.redistributed_visits, cw.attempted_draws, cw.accepted_draws)
    a, w = ctx.download_accum(0)
    ra, rw = whole.download_accum(0)
    m = ra != 0
    assert np.array_equal(a != 0, m)
    assert float(np.max(np.abs(a[m] - ra[m]) / np.abs(ra[m]))) < TOL
    assert float(np.max(np.abs(w[rw != 0] - rw[rw != 0]) / rw[rw != 0])) < TOL


def test_upload_stream_error_paths(gpu_ctx_factory):
    ctx = gpu_ctx_factory()
    part = _abi.Visits()
    with pytest.raises(capi.LentilError):
        ctx.visits_append(part)                      # no begin
    with pytest.raises(capi.LentilError):
        ctx.visits_end()
    lay = _abi.Visits()
    lay.visits_per_pixel, lay.pixels_per_row, lay.pixel_row_stride = 9, 0, 1
    with pytest.raises(capi.LentilError):
        ctx.visits_begin(lay)                        # pixels_per_row == 0
    lay.pixels_per_row = 8
    ctx.visits_begin(lay)
    part.n, part.n_extra = 4, 1
    with pytest.raises(capi.LentilError):
        ctx.visits_append(part)                      # other column count
    part.n_extra = 0
    with pytest.raises(capi.LentilError):
        ctx.visits_append(part)                      # null columns
    with pytest.raises(capi.LentilError):
        ctx.visits_wait(12345)                       # unknown ticket
    assert ctx.visits_end() == 0                     # an empty stream is a stream
