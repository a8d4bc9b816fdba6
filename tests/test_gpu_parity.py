"""GPU parity: the HIP path (through the C-ABI) against the oracle on identical inputs.

Bars (BASELINE.json north_star): accepted-draw lists (visit, attempt, pixel) bit-identical;
accumulated radiance / weights within 1e-5 relative (fp32 atomic summation order differs from the
sequential reference).  The oracle also keeps an fp64 shadow accumulation, so the error of each
side against the exact sum can be told apart.
"""
import ctypes as C
import os

import numpy as np
import pytest

import common
import oracle_lib
from pota_amd import _abi, bokeh, capi

pytestmark = pytest.mark.gpu

TOL = 1e-5


def gpu_run(ctx, p, table, visits, n_aovs=1, bokeh_tables=None, log_cap=1 << 22, lens_mode=0, kinds=None, compiled=True):
    ctx.set_params(p)
    ctx.set_lens_mode(lens_mode)
    if table is not None:
        ctx.set_lens(table)
        assert ctx.lens_is_compiled() or not compiled       # the two benchmark lenses have a compiled-in kernel
    ctx.set_bokeh(bokeh_tables)
    ctx.alloc_frame(n_aovs, kinds)
    ctx.set_draw_log(log_cap)
    ctx.upload_visits(visits)
    ctx.clear_frame()
    ctx.redistribute()
    ctx.resolve()
    ctx.sync()
    c = ctx.counters()
    assert c.worklist_overflow == 0
    return c


def check_frame(ctx, ref, n_aovs=1, tol=TOL, kinds=None, stats=None):
    """accumulators + weight + resolved image vs oracle; returns the worst relative error.
    stats (a dict, optional) receives the two figures a report should carry beside it: `vs_fp32`, the worst
    |gpu - oracle_fp32| / |oracle_fp32| over buffers and resolved images -- the reference CPU imager's own fp32 numbers, what
    BASELINE.json's north_star names -- and `fp32_own`, how far those fp32 numbers are themselves from the exact sums.

    Every value is compared; the fp64 arithmetic runs only over the values that are not bit for bit the oracle's fp32 ones
    (a pixel no draw reaches holds its own visits' sum in iterator order on both sides: equal, and then within any bar the
    oracle's own fp32 value is within).  A 4K frame with nine AOVs is 300 M values; a draw reaches an eighth of them."""
    worst = 0.0
    vs32, own32 = 0.0, 0.0
    rw = ref.weight()
    # (one copy of the pixel records for all AOVs: lentil_hip_download_accum fetches the whole block per call -- nine times
    # 1.3 GB for a 4K frame with nine AOVs)
    rec = ctx.download_records()

    def differing(got, want):
        """flat indices where the two fp32 arrays are not bit for bit equal (NaN counts as different)"""
        return np.flatnonzero(got.ravel() != want.ravel())

    w64 = [None]

    def weight64():
        if w64[0] is None:
            w64[0] = ref.weight64()
        return w64[0]

    for a in range(n_aovs):
        buf, w = rec[:, 4 * a:4 * a + 4], rec[:, 4 * n_aovs]
        rb = ref.buffer(a)
        e64 = [None]

        def exact64():
            if e64[0] is None:
                e64[0] = ref.buffer64(a).ravel()
            return e64[0]

        assert np.array_equal(buf == 0, rb == 0) or np.allclose(buf.ravel()[exact64() == 0], 0, atol=1e-30)
        ne = differing(np.ascontiguousarray(buf), rb)
        if ne.size:
            # relative to the exact (fp64) sum; every contribution on this path is non-negative
            exact = exact64()[ne]
            got = np.ascontiguousarray(buf).ravel()[ne].astype(np.float64)
            r32 = rb.ravel()[ne].astype(np.float64)
            m = exact != 0
            assert bool(np.all(got[~m] == 0)) or np.allclose(got[~m], 0, atol=1e-30)
            if m.any():
                e = float(np.max(np.abs(got[m] - exact[m]) / np.abs(exact[m])))
                worst = max(worst, e)
                # ... and against the reference's own fp32 buffer (what BASELINE.json's north_star names): within the tolerance
                # plus what that buffer's sequential fp32 sum is itself off the exact one (pixels on which thousands of draws
                # pile up: up to 1.3e-5)
                d32 = np.abs(got[m] - r32[m])
                own = np.abs(r32[m] - exact[m])
                nz = np.abs(r32[m]) > 0
                if nz.any():
                    vs32 = max(vs32, float(np.max(d32[nz] / np.abs(r32[m])[nz])))
                own32 = max(own32, float(np.max(own / np.abs(exact[m]))))
                assert bool(np.all(d32 <= tol * np.abs(exact[m]) + own)), "fp32 buffers: worst excess %.3e" % float(
                    np.max((d32 - own) / np.abs(exact[m])))
        if a == 0:
            assert np.array_equal(w != 0, rw != 0)
            nw = differing(np.ascontiguousarray(w), rw)
            if nw.size:
                rw64 = weight64()[nw]
                ew = float(np.max(np.abs(np.ascontiguousarray(w)[nw].astype(np.float64) - rw64) / rw64))
                worst = max(worst, ew)
        img = ctx.download_aov(a)
        rimg = ref.resolve(a)
        ni = differing(img, rimg)
        if ni.size:
            # the resolved image: like the buffers -- against the exact quotient where the AOV is a weighted sum (a closest
            # AOV is a copy), and against the reference's fp32 image within what that is itself off
            gi = img.ravel()[ni].astype(np.float64)
            ri = rimg.ravel()[ni].astype(np.float64)
            xi = ri.copy()
            if kinds is None or kinds[a] == 0:
                wz = weight64()[ni // 4]
                ex = exact64()[ni]
                xi = np.where(wz != 0, ex / np.where(wz != 0, wz, 1.0), ex)
            mi = ri != 0
            assert bool(np.all(gi[~mi] == 0)) or np.allclose(gi[~mi], 0, atol=1e-30), "the images' zero patterns differ"
            if mi.any():
                own_i = np.abs(ri[mi] - xi[mi])
                di = np.abs(gi[mi] - ri[mi])
                vs32 = max(vs32, float(np.max(di / np.abs(ri[mi]))))
                own32 = max(own32, float(np.max(own_i / np.abs(ri[mi]))))
                ei = float(np.max(np.abs(gi[mi] - xi[mi]) / np.abs(ri[mi])))
                worst = max(worst, ei)
                assert bool(np.all(di <= tol * np.abs(ri[mi]) + own_i)), "fp32 image: worst excess %.3e" % float(
                    np.max((di - own_i) / np.abs(ri[mi])))
    if stats is not None:
        stats["vs_fp32"] = max(stats.get("vs_fp32", 0.0), vs32)
        stats["fp32_own"] = max(stats.get("fp32_own", 0.0), own32)
    assert worst < tol, "max relative error %.3e" % worst
    return worst


def check_logs(ctx, ref):
    g = common.sort_log(ctx.draw_log())
    r = common.sort_log(ref.log())
    assert g.shape == r.shape, "accepted draws: gpu %d vs oracle %d" % (g.shape[0], r.shape[0])
    assert np.array_equal(g, r), "accepted-draw (visit, attempt, pixel) lists differ"


# --------------------------------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------------------------------
def test_aperture_sample_disk_bit_exact(orc, gpu_ctx_factory):
    p, model, table, keep = common.po_setup(64, 48)
    ctx = gpu_ctx_factory()
    ctx.set_params(p)
    rng = np.random.default_rng(1)
    a = rng.integers(0, 2 ** 32, 20000, dtype=np.uint64).astype(np.uint32)
    b = rng.integers(0, 4000, 20000, dtype=np.uint64).astype(np.uint32)
    got = ctx.test_aperture_sample(a, b)
    exp = np.empty_like(got)
    tmp = (C.c_double * 2)()
    for i in range(a.shape[0]):
        orc.orc_po_aperture_sample(C.byref(p), None, int(a[i]), int(b[i]), tmp)
        exp[i] = tmp[0], tmp[1]
    assert np.array_equal(got, exp)


def test_lt_sample_aperture_bit_exact(orc, gpu_ctx_factory):
    p, model, table, keep = common.po_setup(64, 48)
    ctx = gpu_ctx_factory()
    ctx.set_params(p)
    ctx.set_lens(table)
    rng = np.random.default_rng(2)
    n = 4096
    scene = np.stack([rng.uniform(-4000, 4000, n), rng.uniform(-3000, 3000, n), rng.uniform(400, 20000, n)], 1)
    ap = rng.uniform(-1, 1, (n, 2)) * p.aperture_radius
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
    assert (eT > 0).mean() > 0.3
    # NaN-aware exact comparison
    assert np.array_equal(T, eT, equal_nan=True)
    assert np.array_equal(sensor, es, equal_nan=True)
    assert np.array_equal(out, eo, equal_nan=True)


def test_trace_bw_po_bit_exact(orc, gpu_ctx_factory):
    p, model, table, keep = common.po_setup(64, 48)
    ctx = gpu_ctx_factory()
    ctx.set_params(p)
    ctx.set_lens(table)
    rng = np.random.default_rng(3)
    n = 4096
    target = np.stack([rng.uniform(-600, 600, n), rng.uniform(-400, 400, n), rng.uniform(500, 5000, n)], 1)
    px = rng.integers(0, 64, n).astype(np.int32)
    py = rng.integers(0, 48, n).astype(np.int32)
    att = rng.integers(0, 3000, n).astype(np.int32)
    xy, ok = ctx.test_trace_bw_po(target, px, py, att)
    lens = orc.orc_lens_create(C.byref(table))
    exy, eok = np.zeros_like(xy), np.zeros_like(ok)
    sp = (C.c_double * 2)()
    for i in range(n):
        eok[i] = orc.orc_trace_ray_bw_po(C.byref(p), lens, None, oracle_lib.darr(*target[i]), sp, int(px[i]),
                                         int(py[i]), int(att[i]), p.lambda_bw, None)
        if eok[i]:
            exy[i] = sp[0], sp[1]
    orc.orc_lens_destroy(lens)
    assert np.array_equal(ok, eok)
    assert np.array_equal(xy[ok == 1], exy[ok == 1])


# --------------------------------------------------------------------------------------------------
# whole path
# --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("lens_mode", [0, 1], ids=["compiled", "tables"])
@pytest.mark.parametrize("override", [0, 64])
def test_po_redistribute_parity(orc, gpu_ctx_factory, override, lens_mode):
    """compiled = straight-line generated kernel of the shipped lens; tables = LDS table interpreter."""
    W, H, M = 96, 64, 9
    p, model, table, keep = common.po_setup(W, H, samples_override=override)
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.02)
    ref = common.run_oracle(orc, p, table, visits)
    ctx = gpu_ctx_factory()
    c = gpu_run(ctx, p, table, visits, lens_mode=lens_mode)
    rc = ref.counters()
    assert (c.redistributed_visits, c.attempted_draws, c.accepted_draws) == (
        rc.redistributed_visits, rc.attempted_draws, rc.accepted_draws)
    check_logs(ctx, ref)
    check_frame(ctx, ref)


def test_direct_accumulation_is_bit_exact(orc, gpu_ctx_factory):
    """No highlights: every pixel only receives its own visits, summed in iterator order -> the
    accumulators must equal the sequential reference bit for bit."""
    W, H, M = 80, 40, 9
    p, model, table, keep = common.po_setup(W, H)
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.0, n_extra=2)
    ref = common.run_oracle(orc, p, table, visits, n_aovs=3)
    ctx = gpu_ctx_factory()
    c = gpu_run(ctx, p, table, visits, n_aovs=3)
    assert c.redistributed_visits == 0
    for a in range(3):
        buf, w = ctx.download_accum(a)
        assert np.array_equal(buf, ref.buffer(a))
        assert np.array_equal(w, ref.weight())
        assert np.array_equal(ctx.download_aov(a), ref.resolve(a))


def test_po_extra_aovs_and_m36(orc, gpu_ctx_factory):
    """K = 3 extra AOVs, 36 visits per pixel (AA 4, filter width 1.5 -> inv density 1/16)."""
    W, H, M = 40, 30, 36
    p, model, table, keep = common.po_setup(W, H, aa=4, filter_width=1.5, samples_override=32)
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.01, n_extra=3)
    ref = common.run_oracle(orc, p, table, visits, n_aovs=4)
    ctx = gpu_ctx_factory()
    gpu_run(ctx, p, table, visits, n_aovs=4)
    check_logs(ctx, ref)
    check_frame(ctx, ref, n_aovs=4)


def test_petzval_8_aovs(orc, gpu_ctx_factory):
    """BASELINE config 4 shape at test size: petzval-class table (degree-9 terms, heavy vignetting ->
    many failing draws and several acceptance rounds), beauty + 8 AOVs."""
    W, H, M = 48, 32, 9
    p, model, table, keep = common.po_setup(W, H, lens="petzval_58mm", samples_override=32)
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.02, n_extra=8)
    ref = common.run_oracle(orc, p, table, visits, n_aovs=9)
    ctx = gpu_ctx_factory()
    c = gpu_run(ctx, p, table, visits, n_aovs=9)
    rc = ref.counters()
    assert (c.attempted_draws, c.accepted_draws) == (rc.attempted_draws, rc.accepted_draws)
    assert rc.attempted_draws > 1.2 * rc.accepted_draws        # the failure paths are exercised
    check_logs(ctx, ref)
    check_frame(ctx, ref, n_aovs=9)


@pytest.mark.parametrize("n_aovs", [15, 16])
def test_the_widest_records(orc, gpu_ctx_factory, n_aovs):
    """LENTIL_MAX_AOVS gaussian AOVs: an accepted draw adds 4 * 16 + 1 = 65 floats, one more than a wave has lanes --
    the wide accept's (draw, float) lane layout has no room for that record and the narrow form takes it (round-4
    ADVICE: with 16 all-gaussian AOVs no splat was issued).  15 AOVs (61 floats) is the widest record the wide form serves."""
    W, H, M = 40, 28, 9
    p, model, table, keep = common.po_setup(W, H, samples_override=40)
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.02, n_extra=n_aovs - 1)
    ref = common.run_oracle(orc, p, table, visits, n_aovs=n_aovs)
    ctx = gpu_ctx_factory()
    c = gpu_run(ctx, p, table, visits, n_aovs=n_aovs)
    rc = ref.counters()
    assert rc.accepted_draws > 0
    assert (c.attempted_draws, c.accepted_draws) == (rc.attempted_draws, rc.accepted_draws)
    check_logs(ctx, ref)
    check_frame(ctx, ref, n_aovs=n_aovs)


@pytest.mark.parametrize("ragged", [False, True])
def test_closest_filter_aovs(orc, gpu_ctx_factory, ragged):
    """closest-original AOVs (e.g. P, N, Z; src/lentil.h:832-837): per pixel the candidate with the
    smallest |Z| wins, later visits win ties; resolved as (r, g, b, 1).  Mixed with gaussian AOVs."""
    W, H, M = 64, 40, 9
    kinds = [0, 1, 0, 1]
    p, model, table, keep = common.po_setup(W, H, samples_override=48)
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.03, n_extra=3)
    if ragged:
        n = cols["rgba"].shape[0]
        rng = np.random.default_rng(11)
        cols["pixel"] = (rng.integers(0, W, n) | (rng.integers(0, H, n) << 16)).astype(np.uint32)
        visits, keepv = capi.make_visits(cols, visits_per_pixel=0)
    lens = orc.orc_lens_create(C.byref(table))
    ref = oracle_lib.Frame(orc, p, n_aovs=4, kinds=kinds, keep_log=True)
    ref.run_auto(lens, None, visits)
    orc.orc_lens_destroy(lens)
    ctx = gpu_ctx_factory()
    gpu_run(ctx, p, table, visits, n_aovs=4, kinds=kinds)
    check_logs(ctx, ref)
    for a in (1, 3):                      # closest: exact
        buf, _ = ctx.download_accum(a)
        assert np.array_equal(buf, ref.buffer(a))
        img = ctx.download_aov(a)
        assert np.array_equal(img, ref.resolve(a))
        assert np.all(img[buf.any(axis=1), 3] == 1.0)
    for a in (0, 2):                      # gaussian: tolerance
        buf, w = ctx.download_accum(a)
        exact = ref.buffer64(a)
        m = exact != 0
        assert float(np.max(np.abs(buf[m] - exact[m]) / np.abs(exact[m]))) < TOL


class _InProcessDist:
    """Stands in for torch.distributed between HipEngines living in this process (one GPU, one thread per
    "rank"): all_reduce / all_gather / batch_isend_irecv with the call signatures frame_step and
    frame_step_bands use.  Lets the multi-rank paths run on a single MI355X."""

    class ReduceOp:
        SUM, MIN = "sum", "min"

    class _Req:
        def wait(self):
            return None

    class _Shared:
        def __init__(self, world):
            import threading
            self.world = world
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world
            import queue
            self.mail = {(a, b): queue.Queue() for a in range(world) for b in range(world)}

    isend, irecv = "isend", "irecv"

    @staticmethod
    def P2POp(op, tensor, peer):
        return (op, tensor, peer)

    def __init__(self, shared, rank):
        self.sh, self.rank = shared, rank

    def is_initialized(self):
        return True

    def get_world_size(self):
        return self.sh.world

    def get_rank(self):
        return self.rank

    def _sync(self):
        import torch
        torch.cuda.synchronize()
        self.sh.barrier.wait()

    def all_reduce(self, t, op):
        import torch
        self.sh.slots[self.rank] = t
        self._sync()
        if self.rank == 0:
            r = self.sh.slots[0].clone()
            for o in self.sh.slots[1:]:
                r = torch.minimum(r, o) if op == "min" else r + o
            for o in self.sh.slots:
                o.copy_(r)
        self._sync()

    def all_gather(self, out, t):
        self.sh.slots[self.rank] = t.clone()
        self._sync()
        for k in range(self.sh.world):
            out[k].copy_(self.sh.slots[k])
        self._sync()

    def batch_isend_irecv(self, ops):
        # sends and receives pair up per (source, destination) in posting order and involve nobody else, as with a
        # real backend: a rank with nothing to exchange does not take part (a barrier here would stall on it)
        import torch
        for op, t, peer in ops:
            if op == "isend":
                c = t.clone()
                torch.cuda.synchronize()
                self.sh.mail[(self.rank, peer)].put(c)
        for op, t, peer in ops:
            if op == "irecv":
                t.copy_(self.sh.mail[(peer, self.rank)].get(timeout=60))
        torch.cuda.synchronize()
        return [self._Req() for _ in ops]


def _run_ranks(fn, world):
    """fn(rank, dist) on one thread per rank; re-raises the first failure."""
    import threading
    shared = _InProcessDist._Shared(world)
    errors = []

    def run(rank):
        try:
            fn(rank, _InProcessDist(shared, rank))
        except Exception as e:          # pragma: no cover
            errors.append(e)
            shared.barrier.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    if errors:
        raise errors[0]


def _compare_with_whole(ctx, whole, kinds, rows=None):
    """accumulators / resolved AOVs of `ctx` (restricted to `rows` of the frame) against the whole-frame context"""
    p_y, p_x = whole.P.yres, whole.P.xres
    sel = np.zeros(p_y, bool)
    sel[slice(*rows) if rows else slice(None)] = True
    sel = np.repeat(sel, p_x)
    for a, kind in enumerate(kinds):
        buf, w = ctx.download_accum(a)
        ref, rw = whole.download_accum(a)
        img, rimg = ctx.download_aov(a), whole.download_aov(a)
        buf, ref, img, rimg, w, rw = buf[sel], ref[sel], img[sel], rimg[sel], w[sel], rw[sel]
        if kind:                                            # closest: identical winners and values
            assert np.array_equal(buf, ref) and np.array_equal(img, rimg)
            assert np.count_nonzero(ref) > 0
        else:                                               # gaussian: fp32 summation order differs
            m = ref != 0
            assert np.array_equal(buf != 0, m)
            assert float(np.max(np.abs(buf[m] - ref[m]) / np.abs(ref[m]))) < TOL
            assert float(np.max(np.abs(w[rw != 0] - rw[rw != 0]) / rw[rw != 0])) < TOL
            mi = rimg != 0
            assert float(np.max(np.abs(img[mi] - rimg[mi]) / np.abs(rimg[mi]))) < 2 * TOL


def test_two_partitions_on_one_gpu_match_the_whole_frame(orc, gpu_ctx_factory):
    """Multi-GPU logic on one device: rows r mod 2 in two contexts, closest-AOV key
    exchange + owner gather, sum of the accumulators, local resolve -- against one context that
    processes the whole frame (which test_closest_filter_aovs pins to the oracle)."""
    from pota_amd import distributed, workload
    W, H, M = 64, 40, 9
    kinds = [0, 1, 0]
    p, model, table, keep = common.po_setup(W, H, samples_override=48)
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.03, n_extra=2)
    whole = gpu_ctx_factory()
    gpu_run(whole, p, table, visits, n_aovs=3, kinds=kinds)
    whole.P = p

    engines, keepalive = [], []
    for rank in range(2):
        n_local = workload.frame_visit_count(W, H, M, 2, rank)
        c = workload.generate(np, 0, n_local, W, H, M, f_hi=0.03, focus_dist=150.0,
                              tan_half_fov=common.tan_half_fov(p), row_stride=2, row_offset=rank, n_extra=2)
        v, kv = capi.make_visits(c, visits_per_pixel=M, pixels_per_row=W, pixel_y0=rank, pixel_row_stride=2)
        ctx = gpu_ctx_factory()
        ctx.set_params(p); ctx.set_lens(table); ctx.set_bokeh(None)
        ctx.alloc_frame(3, kinds)
        ctx.upload_visits(v)
        keepalive.append((c, v, kv))
        engines.append(distributed.HipEngine(ctx, rows=p.yres))
    assert engines[0].zkey is not None

    def step(rank, dist):
        distributed.frame_step(engines[rank], dist)
        engines[rank].ctx.sync()

    _run_ranks(step, 2)
    for rank in range(2):
        _compare_with_whole(engines[rank].ctx, whole, kinds)


@pytest.mark.parametrize("world,bounds,f_hi", [(2, None, 0.03), (3, None, 0.03), (3, [0, 9, 31, 45], 0.03), (3, None, 0.0015)],
                         ids=["2", "3", "3-unequal", "3-sparse"])
def test_tiled_output_on_one_gpu_matches_the_whole_frame(orc, gpu_ctx_factory, monkeypatch, world, bounds, f_hi):
    """Tiled multi-GPU mode on one device: each "rank" processes a band of rows, tells the others which rows
    it touched, sends the foreign ones to their owners (lentil_hip_merge_rows) and resolves its band
    (lentil_hip_resolve_rows).  Every band must equal the same rows of a whole-frame context; a second pass
    checks that the row-limited clear leaves nothing behind.  "unequal": bands cut the way
    distributed.rebalance() cuts them when the ranks' pass times differ.  "sparse": few highlights, so the rows a
    band touches in its neighbours are mostly empty and travel as lists of pixels (lentil_hip_compact_rows /
    _merge_sparse) instead of whole rows."""
    from pota_amd import distributed, workload
    if bounds is not None:
        monkeypatch.setattr(distributed, "SPARSE_EXCHANGE", False)      # this case: whole (packed) rows
    W, H, M = 64, 45, 9
    kinds = [0, 1, 0]
    p, model, table, keep = common.po_setup(W, H, samples_override=48)
    visits, cols = common.make_stream(p, W, H, M, f_hi=f_hi, n_extra=2)
    whole = gpu_ctx_factory()
    gpu_run(whole, p, table, visits, n_aovs=3, kinds=kinds)
    whole.P = p

    engines, keepalive, bands = [], [], []
    for rank in range(world):
        b_lo, b_hi = distributed.band_of(rank, world, H, p.yres, bounds)
        v_hi = min(b_hi, H)
        c = workload.generate(np, b_lo * W * M, v_hi * W * M, W, H, M, f_hi=f_hi, focus_dist=150.0,
                              tan_half_fov=common.tan_half_fov(p), n_extra=2)
        v, kv = capi.make_visits(c, visits_per_pixel=M, pixels_per_row=W, pixel_y0=b_lo)
        ctx = gpu_ctx_factory()
        ctx.set_params(p); ctx.set_lens(table); ctx.set_bokeh(None)
        ctx.alloc_frame(3, kinds)
        ctx.upload_visits(v)
        keepalive.append((c, v, kv))
        engines.append(distributed.HipEngine(ctx, rows=p.yres))
        bands.append((b_lo, b_hi))
    assert bands[0][0] == 0 and bands[-1][1] == p.yres

    got = {}

    def step(rank, dist):
        for _ in range(2):                                   # second pass: clear_frame wipes only the touched rows
            got[rank] = distributed.frame_step_bands(engines[rank], dist, H, p.yres, bounds)
            engines[rank].ctx.sync()

    _run_ranks(step, world)
    reach = 0
    for rank in range(world):
        assert got[rank] == bands[rank]
        lo, hi = engines[rank].ctx.touched_rows()
        reach = max(reach, bands[rank][0] - lo, hi - bands[rank][1])
        _compare_with_whole(engines[rank].ctx, whole, kinds, rows=bands[rank])
    assert reach > 0            # draws did cross the band boundaries
    forms = [f for r in range(world) for f in distributed.LAST_FORMS[r]]
    if bounds is not None:
        assert any(f < 0 for f in forms) and not any(f > 0 for f in forms)     # whole (packed) rows only
    else:
        assert any(f > 0 for f in forms) and not any(f < 0 for f in forms)     # pixel lists only



def _tiled_rank_process(rank, world, port, W, H, M, f_hi, kinds, bounds, sparse, q):
    """One rank of test_tiled_output_two_processes: a process of its own, torch.distributed (gloo) between the ranks,
    all of them on GPU 0."""
    import os
    import sys
    sys.path.insert(0, common.ROOT)
    sys.path.insert(0, os.path.join(common.ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from pota_amd import distributed, workload
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    distributed.SPARSE_EXCHANGE = sparse
    p, model, table, keep = common.po_setup(W, H, samples_override=48)
    b_lo, b_hi = distributed.band_of(rank, world, H, p.yres, bounds)
    c = workload.generate(np, b_lo * W * M, min(b_hi, H) * W * M, W, H, M, f_hi=f_hi, focus_dist=150.0,
                          tan_half_fov=common.tan_half_fov(p), n_extra=len(kinds) - 1)
    v, kv = capi.make_visits(c, visits_per_pixel=M, pixels_per_row=W, pixel_y0=b_lo)
    ctx = capi.Context(0)
    ctx.set_params(p); ctx.set_lens(table); ctx.set_bokeh(None)
    ctx.alloc_frame(len(kinds), kinds)
    ctx.upload_visits(v)
    eng = distributed.HipEngine(ctx, rows=p.yres)
    for _ in range(2):                                       # second pass: blind, row-limited clear
        band = distributed.frame_step_bands(eng, dist, H, p.yres, bounds)
        ctx.sync()
    out = {"band": band, "forms": distributed.LAST_FORMS[rank]}
    for a in range(len(kinds)):
        buf, w = ctx.download_accum(a)
        out[a] = (buf, w, ctx.download_aov(a))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()


@pytest.mark.parametrize("sparse", [True, False], ids=["pixel-lists", "rows"])
def test_tiled_output_two_processes(orc, gpu_ctx_factory, sparse):
    """The tiled step as bench.py --gpus N runs it -- one process per rank, torch.distributed collectives and
    batch_isend_irecv between them (gloo here, both ranks on the one GPU of the test box; RCCL in the real thing) --
    against a context that has the whole frame.  Unequal bands, closest-filtered AOV included."""
    import socket
    import torch.multiprocessing as mp
    W, H, M = 64, 45, 9
    kinds = [0, 1, 0]
    bounds = [0, 19, 45]
    f_hi = 0.0015 if sparse else 0.03
    p, model, table, keep = common.po_setup(W, H, samples_override=48)
    visits, cols = common.make_stream(p, W, H, M, f_hi=f_hi, n_extra=2)
    whole = gpu_ctx_factory()
    gpu_run(whole, p, table, visits, n_aovs=3, kinds=kinds)
    ref = {a: (whole.download_accum(a), whole.download_aov(a)) for a in range(3)}
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_tiled_rank_process, args=(r, 2, port, W, H, M, f_hi, kinds, bounds, sparse, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    forms = got[0]["forms"] + got[1]["forms"]
    assert any(f > 0 for f in forms) if sparse else any(f < 0 for f in forms)
    for rank in range(2):
        lo, hi = got[rank]["band"]
        sel = np.zeros(p.yres, bool); sel[lo:hi] = True
        sel = np.repeat(sel, p.xres)
        for a, kind in enumerate(kinds):
            buf, w, img = got[rank][a]
            (rbuf, rw), rimg = ref[a]
            buf, w, img, rbuf, rw, rimg = buf[sel], w[sel], img[sel], rbuf[sel], rw[sel], rimg[sel]
            if kind:
                assert np.array_equal(buf, rbuf) and np.array_equal(img, rimg)
            else:
                m = rbuf != 0
                assert np.array_equal(buf != 0, m)
                assert float(np.max(np.abs(buf[m] - rbuf[m]) / np.abs(rbuf[m]))) < TOL
                assert float(np.max(np.abs(w[rw != 0] - rw[rw != 0]) / rw[rw != 0])) < TOL
                mi = rimg != 0
                assert float(np.max(np.abs(img[mi] - rimg[mi]) / np.abs(rimg[mi]))) < 2 * TOL


@pytest.mark.parametrize("chroma,lens_mode,override", [(0.5, 0, 48), (0.5, 1, 48), (1.0, 0, 0), (-0.5, 0, 48)])
def test_po_chromatic_aberration(orc, gpu_ctx_factory, chroma, lens_mode, override):
    """abb_chromatic != 0 in polynomial-optics mode (src/lentil_filter.cpp:248-299): three wavelength
    channels per attempt, each failing channel takes one off the draw count, channel c feeds colour
    component c three-fold.  Accepted (attempt, channel, pixel) lists bit-identical; a gaussian and a
    closest extra AOV ride along.  abb_chromatic < 0 runs three white channels at 0.55 (upstream quirk)."""
    W, H, M = 64, 48, 9
    kinds = [0, 0, 1]
    p, model, table, keep = common.po_setup(W, H, samples_override=override, abb_chromatic=chroma)
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.03, n_extra=2)
    lens = orc.orc_lens_create(C.byref(table))
    ref = oracle_lib.Frame(orc, p, n_aovs=3, kinds=kinds, keep_log=True)
    ref.run_auto(lens, None, visits)
    orc.orc_lens_destroy(lens)
    rc = ref.counters()
    assert rc.accepted_draws > 2 * rc.attempted_draws            # nearly three splats per attempt
    ctx = gpu_ctx_factory()
    c = gpu_run(ctx, p, table, visits, n_aovs=3, kinds=kinds, lens_mode=lens_mode)
    assert (c.redistributed_visits, c.attempted_draws, c.accepted_draws) == \
        (rc.redistributed_visits, rc.attempted_draws, rc.accepted_draws)
    check_logs(ctx, ref)
    chan = ctx.draw_log()[:, 1] >> 30
    assert set(np.unique(chan)) == {0, 1, 2}
    # three splats per attempt: about three times the fp32 additions per pixel of the other cases, in an
    # order the atomics choose; worst pixel measured 1.05e-5 from the exact (fp64) sum in some runs, so
    # this mode asserts 2e-5 (the index lists above are exact; SURVEY App. C.6 keeps abb_chromatic out of
    # the 1e-5 fixtures)
    tol = 2 * TOL
    for a in (0, 1):
        buf, w = ctx.download_accum(a)
        exact = ref.buffer64(a)
        m = exact != 0
        assert np.array_equal(buf != 0, ref.buffer(a) != 0)
        assert float(np.max(np.abs(buf[m] - exact[m]) / np.abs(exact[m]))) < tol
        # resolved image against the exact quotient: numerator and denominator are each within TOL of
        # their exact sums (asserted above / below), the quotient of the two within 2 TOL
        w64 = ref.weight64()
        eimg = np.where(w64[:, None] != 0, exact / np.where(w64 != 0, w64, 1.0)[:, None], exact)
        img = ctx.download_aov(a)
        mi = eimg != 0
        assert float(np.max(np.abs(img[mi] - eimg[mi]) / np.abs(eimg[mi]))) < 2 * tol
    w64 = ref.weight64(); mw = w64 != 0
    assert float(np.max(np.abs(ctx.download_accum(0)[1][mw] - w64[mw]) / w64[mw])) < tol
    buf, _ = ctx.download_accum(2)                                # closest AOV: exact
    assert np.array_equal(buf, ref.buffer(2))
    ref.close()


def test_lentil_debug_aov(orc, gpu_ctx_factory):
    """The lentil_debug AOV (src/lentil_filter.cpp:209-211, src/lentil.h:838-845): value = draw count of the
    visit, written by redistributed draws only, closest through its own z-buffer; no visit column."""
    W, H, M = 64, 40, 9
    kinds = [_abi.FILTER_GAUSSIAN, _abi.FILTER_CLOSEST_DEBUG, _abi.FILTER_CLOSEST]
    p, model, table, keep = common.po_setup(W, H, samples_override=0)          # draw counts from the CoC formula
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.03, n_extra=2)
    cols["extra"][0] = None
    visits, keepv = capi.make_visits(cols, visits_per_pixel=M, pixels_per_row=W)
    # the oracle reads every column: hand it a dummy one for the debug AOV (its values are ignored)
    ocols = dict(cols); ocols["extra"] = [np.zeros_like(cols["rgba"]), cols["extra"][1]]
    ovisits, okeep = capi.make_visits(ocols, visits_per_pixel=M, pixels_per_row=W)
    lens = orc.orc_lens_create(C.byref(table))
    ref = oracle_lib.Frame(orc, p, n_aovs=3, kinds=kinds, keep_log=True)
    ref.run(lens, None, ovisits)
    orc.orc_lens_destroy(lens)
    ctx = gpu_ctx_factory()
    gpu_run(ctx, p, table, visits, n_aovs=3, kinds=kinds)
    check_logs(ctx, ref)
    dbg, _ = ctx.download_accum(1)
    rdbg = ref.buffer(1)
    assert np.array_equal(dbg, rdbg)
    counts = np.unique(rdbg[:, 0])
    assert len(counts) > 3 and counts.max() > 4                    # several different draw counts landed
    assert np.array_equal(ctx.download_aov(1), ref.resolve(1))     # (count, count, count, 1)
    # pixels no draw reached stay empty although their own visits wrote the ordinary closest AOV
    own, _ = ctx.download_accum(2)
    assert np.array_equal(own, ref.buffer(2))
    assert ((rdbg[:, 0] == 0) & own.any(axis=1)).sum() > 0
    ref.close()


@pytest.mark.parametrize("layout", ["uniform", "ragged_runs", "ragged_atomics"])
def test_lentil_debug_visits_inside_the_lens(orc, gpu_ctx_factory, monkeypatch, layout):
    """src/lentil_filter.cpp:209-212 takes lentil_debug's value, samples * redistribute, BEFORE :240 clears redistribute for
    a sample inside the lens: such a visit reaches the direct path (src/lentil.h:938-955) with a non-zero count and competes
    for the debug z-buffer (src/lentil.h:838-845) at its own pixel.  A stream with a few hundred visits at 5 % of the lens
    length, through every scan kernel that serves frames with closest AOVs, against the oracle."""
    if layout == "ragged_atomics":
        monkeypatch.setenv("LENTIL_SCAN_RUNS", "0")
    W, H, M = 64, 40, 9
    kinds = [_abi.FILTER_GAUSSIAN, _abi.FILTER_CLOSEST_DEBUG, _abi.FILTER_CLOSEST]
    p, model, table, keep = common.po_setup(W, H, samples_override=0)
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.03, n_extra=2)
    n = cols["rgba"].shape[0]
    rng = np.random.default_rng(23)
    inside = rng.choice(n, 400, replace=False)
    z_in = np.float32(float(table.lens_length) * 0.1 * 0.5)           # camera space is cm, the lens length mm
    cols["pos_z"][inside, 0] = rng.uniform(-0.2, 0.2, inside.size).astype(np.float32) * z_in
    cols["pos_z"][inside, 1] = rng.uniform(-0.2, 0.2, inside.size).astype(np.float32) * z_in
    cols["pos_z"][inside, 2] = -z_in * rng.uniform(0.2, 1.9, inside.size).astype(np.float32)
    cols["pos_z"][inside, 3] = -cols["pos_z"][inside, 2]
    cols["extra"][0] = None
    if layout != "uniform":
        pix = np.arange(n, dtype=np.uint64) // M
        cols["pixel"] = ((pix % W).astype(np.uint32) | ((pix // W).astype(np.uint32) << 16)).astype(np.uint32)
    mk = dict(visits_per_pixel=M, pixels_per_row=W) if layout == "uniform" else dict(visits_per_pixel=0)
    visits, keepv = capi.make_visits(cols, **mk)
    ocols = dict(cols); ocols["extra"] = [np.zeros_like(cols["rgba"]), cols["extra"][1]]
    ovisits, okeep = capi.make_visits(ocols, **mk)
    lens = orc.orc_lens_create(C.byref(table))
    ref = oracle_lib.Frame(orc, p, n_aovs=3, kinds=kinds, keep_log=True)
    ref.run(lens, None, ovisits)
    orc.orc_lens_destroy(lens)
    ctx = gpu_ctx_factory()
    gpu_run(ctx, p, table, visits, n_aovs=3, kinds=kinds)
    check_logs(ctx, ref)
    dbg, _ = ctx.download_accum(1)
    rdbg = ref.buffer(1)
    # the case exists in this stream: a pixel whose debug value comes from one of its OWN visits, not from a draw
    drawn = np.zeros(rdbg.shape[0], bool)
    drawn[ref.log()[:, 2]] = True
    assert ((rdbg[:, 0] != 0) & ~drawn).sum() > 50
    assert np.array_equal(dbg, rdbg)
    assert np.array_equal(ctx.download_aov(1), ref.resolve(1))
    own, _ = ctx.download_accum(2)
    assert np.array_equal(own, ref.buffer(2))
    ref.close()


def test_po_ragged_pixels_and_inv_density(orc, gpu_ctx_factory):
    """Explicit per-visit pixel + per-visit inverse density (ragged footprints / adaptive sampling)."""
    W, H, M = 48, 32, 9
    p, model, table, keep = common.po_setup(W, H, samples_override=16)
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.03)
    n = cols["rgba"].shape[0]
    rng = np.random.default_rng(7)
    px = rng.integers(0, W, n).astype(np.uint32)
    py = rng.integers(0, H, n).astype(np.uint32)
    cols["pixel"] = (px | (py << 16)).astype(np.uint32)
    cols["inv_density"] = rng.choice(np.array([1 / 9., 1 / 16., 0.14], np.float32), n).astype(np.float32)
    visits, keepv = capi.make_visits(cols, visits_per_pixel=0)
    ref = common.run_oracle(orc, p, table, visits)
    ctx = gpu_ctx_factory()
    gpu_run(ctx, p, table, visits)
    check_logs(ctx, ref)
    check_frame(ctx, ref)


def test_po_bokeh_image(orc, gpu_ctx_factory):
    """imagebokeh CDF aperture draws (config 3's sampler) from the reference's example kernel."""
    import os
    tex = np.load(os.path.join(common.ROOT, "tests", "golden", "example_bokeh_kernel_u8.npy")).astype(np.float32) / np.float32(255)
    tables = bokeh.build_tables(tex)
    W, H, M = 64, 48, 9
    p, model, table, keep = common.po_setup(W, H, samples_override=48, bokeh_enable_image=1)
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.02)
    bt = _abi.BokehTable()
    bt.x, bt.y = tables["x"], tables["y"]
    for k in ("cdfRow", "rowIndices", "cdfColumn", "columnIndices"):
        setattr(bt, k, tables[k].ctypes.data)
    ob = orc.orc_bokeh_from_tables(C.byref(bt))
    ref = common.run_oracle(orc, p, table, visits, bokeh=ob)
    ctx = gpu_ctx_factory()
    gpu_run(ctx, p, table, visits, bokeh_tables=tables)
    check_logs(ctx, ref)
    check_frame(ctx, ref)
    orc.orc_bokeh_destroy(ob)


def _poison_stream(cols, n, lens_length_cm, seed):
    """Every branch of the redistribute decision (src/lentil_filter.cpp:105-165,240): on a third of the visits one of
    the special cases, highlights included (their radiance is kept, so the draw path sees them too)."""
    rng = np.random.default_rng(seed)
    kind = rng.integers(0, 24, n)
    pz, vi, tr, rd = cols["pos_z"], cols["volume_ignore"], cols["transmission"], cols["raydir_time"]
    m = kind == 0; vi[m, 0] = 0.5                                   # volume in r
    m = kind == 1; vi[m, 2] = 1e-6                                  # volume in b, tiny but > 0
    m = kind == 2; vi[m, 3] = 1.0                                   # lentil_ignore
    m = kind == 3; tr[m, 1] = 0.25; tr[m, 0] = 0.1                  # transmission
    m = kind == 4; pz[m, 3] = np.float32(1.0e30)                    # Z == AI_INFINITE, raydir from the generator
    m = kind == 5; pz[m, 3] = np.float32(1.0e30); rd[m, :3] = 0.0   # ... and no ray direction
    m = kind == 6; pz[m, :3] = np.float32(5e-5)                     # P ~ 0
    m = kind == 7; pz[m, 2] = -np.float32(lens_length_cm * 0.05)    # inside the lens (PO: |z| < lens_length * 0.1)
    m = (kind == 8) & (rng.integers(0, 16, n) == 0); pz[m, 2] = np.abs(pz[m, 2])   # behind the camera (few: every one
    #                                                                                  of them burns its 5 x samples attempts)
    m = kind == 9; vi[m, 0] = -1.0; vi[m, 3] = -2.0                 # negative flags do not count
    m = kind == 10; tr[m, 3] = 0.9                                  # transmission alpha alone does not count
    rare = rng.integers(0, 8, n) == 0
    m = (kind == 11) & rare; pz[m, 0] = np.float32(np.inf)          # non-finite positions: decided, drawn and rejected
    m = (kind == 12) & rare; pz[m, 2] = np.float32(np.nan)          # the same way on both sides
    return cols


@pytest.mark.parametrize("mode", ["uniform", "extra_aovs", "ragged", "skydome", "bidir_transmission", "add_energy", "metres",
                                  "thinlens"])
def test_redistribute_decision_branches(orc, gpu_ctx_factory, mode):
    """The scan kernels decide with visit_redistributes() -- the tests of visit_prologue() without the draw-count
    arithmetic -- and only flagged visits run the full prologue.  Streams that hit every branch of the decision
    (volume, ignore flag, transmission with and without enable_bidir_transmission, infinite depth with and without
    the skydome, P ~ 0, inside the lens, behind the camera), through all three scan kernels: counters, accepted-draw
    lists and accumulators against the oracle."""
    W, H, M = 64, 40, 9
    kw = {}
    if mode == "skydome":
        kw["enable_skydome"] = 1
    if mode == "bidir_transmission":
        kw["enable_bidir_transmission"] = 1
    if mode == "add_energy":
        kw.update(bidir_add_energy=0.7, bidir_add_energy_minimum_luminance=1.5, bidir_add_energy_transition=60.0)
    if mode == "metres":
        kw["unitModel"] = _abi.UNIT_M
    if mode == "thinlens":
        p, table = common.tl_setup(W, H, samples_override=24), None
    else:
        p, model, table, keep = common.po_setup(W, H, samples_override=24, **kw)
    n_extra = 2 if mode == "extra_aovs" else 0
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.05, n_extra=n_extra)
    n = cols["rgba"].shape[0]
    _poison_stream(cols, n, float(table.lens_length) * 0.1 if table is not None else 1.0, seed=11)
    if mode == "metres":
        cols["pos_z"][:, :3] *= np.float32(0.01)     # the same scene in metres (world_to_camera is the identity)
    if mode == "ragged":
        rng = np.random.default_rng(5)
        cols["pixel"] = (rng.integers(0, W, n).astype(np.uint32) | (rng.integers(0, H, n).astype(np.uint32) << 16)).astype(np.uint32)
        visits, keepv = capi.make_visits(cols, visits_per_pixel=0)
    else:
        visits, keepv = capi.make_visits(cols, visits_per_pixel=M, pixels_per_row=W)
    ref = common.run_oracle(orc, p, table, visits, n_aovs=1 + n_extra)
    rc = ref.counters()
    assert 0 < rc.redistributed_visits < 0.2 * n
    ctx = gpu_ctx_factory()
    c = gpu_run(ctx, p, table, visits, n_aovs=1 + n_extra)
    assert (c.redistributed_visits, c.attempted_draws, c.accepted_draws) == (
        rc.redistributed_visits, rc.attempted_draws, rc.accepted_draws)
    check_logs(ctx, ref)
    # thin lens: the draws of the visits at x = inf all land on pixel (0, 0).  The reference's several thousand
    # sequential fp32 additions end 1.25e-5 off the exact (fp64) sum there; the accept kernel adds the draws of a step
    # that share a pixel as one count x value and stays inside the 1e-5 the checker allows against the exact sum
    check_frame(ctx, ref, n_aovs=1 + n_extra)


@pytest.mark.parametrize("override", [0, 64])
def test_thinlens_redistribute_parity(orc, gpu_ctx_factory, override):
    """BASELINE config 1 shape (thin lens, 64 draws, beauty only) at test size."""
    W, H, M = 96, 64, 9
    p = common.tl_setup(W, H, samples_override=override)
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.02)
    ref = common.run_oracle(orc, p, None, visits)
    ctx = gpu_ctx_factory()
    c = gpu_run(ctx, p, None, visits)
    assert c.accepted_draws == ref.counters().accepted_draws
    check_logs(ctx, ref)
    check_frame(ctx, ref)


@pytest.mark.parametrize("coma", [0.35, 1.0])
def test_thinlens_coma_vignetting_distortion(orc, gpu_ctx_factory, coma):
    """The thin-lens draw's optional branches together (src/lentil_filter.cpp:328-337,379-386,420):
    coma rotation (Eigen AngleAxisd restated), optical vignetting, inverse barrel distortion, hexagonal
    aperture."""
    W, H, M = 96, 64, 9
    p = common.tl_setup(W, H, samples_override=48, abb_coma=coma, optical_vignetting_distance=2.0,
                        optical_vignetting_radius=1.5, abb_distortion=0.15, bokeh_aperture_blades=6)
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.02)
    ref = common.run_oracle(orc, p, None, visits)
    assert ref.counters().accepted_draws > 10000
    ctx = gpu_ctx_factory()
    c = gpu_run(ctx, p, None, visits)
    assert c.accepted_draws == ref.counters().accepted_draws
    assert c.attempted_draws == ref.counters().attempted_draws
    check_logs(ctx, ref)
    check_frame(ctx, ref)


@pytest.mark.parametrize("ctype,coma,vignetting", [(0, 0.0, 0.0), (1, 0.35, 2.0)], ids=["green-magenta", "red-cyan+coma+vignetting"])
def test_thinlens_chromatic_aberration(orc, gpu_ctx_factory, ctype, coma, vignetting):
    """Thin lens with abb_chromatic > 0 (src/lentil_filter.cpp:393-406): every attempt that survives the optical
    vignetting test draws its colour channel from xor128; the channel shifts the focus plane with the distance from
    the frame centre and feeds one colour component three-fold.  The order of the generator's draws is the
    single-threaded one on both sides: accepted (attempt, channel, pixel) lists bit-identical, the generator state
    after the pass identical -- also over a second pass that continues the stream, and from a chosen starting state.
    A gaussian and a closest extra AOV ride along."""
    W, H, M = 96, 64, 9
    kinds = [0, 0, 1]
    p = common.tl_setup(W, H, samples_override=48, abb_chromatic=0.6, abb_chromatic_type=ctype, abb_coma=coma,
                        optical_vignetting_distance=vignetting, optical_vignetting_radius=1.5)
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.02, n_extra=2)
    ctx = gpu_ctx_factory()
    start = None
    for frame in range(3):
        ref = oracle_lib.Frame(orc, p, n_aovs=3, kinds=kinds, keep_log=True)
        if frame == 2:                                   # a starting state of the host's choosing
            start = [0x12345678, 0x9ABCDEF0, 0x0F1E2D3C, 0x4B5A6978]
            ctx.set_xor128_state(start)
        if start is not None:
            orc.orc_frame_set_xor128(ref.h, (C.c_uint32 * 4)(*start))
        ref.run(None, None, visits)
        rc = ref.counters()
        assert rc.redistributed_visits > 100 and rc.accepted_draws > 5000
        c = gpu_run(ctx, p, None, visits, n_aovs=3, kinds=kinds)
        assert (c.redistributed_visits, c.attempted_draws, c.accepted_draws) == \
            (rc.redistributed_visits, rc.attempted_draws, rc.accepted_draws)
        check_logs(ctx, ref)
        chan = ctx.draw_log()[:, 1] >> 30
        assert set(np.unique(chan)) == {0, 1, 2}
        st = (C.c_uint32 * 4)()
        orc.orc_frame_get_xor128(ref.h, st)
        assert ctx.get_xor128_state() == list(st)
        start = list(st)                                 # the next pass continues the stream
        if vignetting:
            assert rc.attempted_draws > rc.accepted_draws        # some attempts were vignetted / left the frame
        check_frame(ctx, ref, n_aovs=3, kinds=kinds)
        ref.close()
    # the three colour components of a highlight's disc differ: the channels are focused at different distances
    img = ctx.download_aov(0).reshape(p.yres, p.xres, 4)
    assert float(np.abs(img[..., 0] - img[..., 2]).max()) > 0.0


def test_sub_batches_when_the_result_pool_is_small(orc, monkeypatch):
    """A chunk whose draws do not fit the result pool is processed in sub-batches of items (and single
    chunk / many chunks give the same answer)."""
    W, H, M = 96, 64, 9
    p, model, table, keep = common.po_setup(W, H, samples_override=64)
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.02)
    ref = common.run_oracle(orc, p, table, visits)
    for chunks, pool in (("1", "40000"), ("5", "20000")):
        monkeypatch.setenv("LENTIL_CHUNKS", chunks)
        monkeypatch.setenv("LENTIL_MAX_POOL_UNITS", pool)
        ctx = capi.Context(0)
        try:
            c = gpu_run(ctx, p, table, visits)
            rc = ref.counters()
            assert (c.attempted_draws, c.accepted_draws) == (rc.attempted_draws, rc.accepted_draws)
            check_logs(ctx, ref)
            check_frame(ctx, ref)
        finally:
            ctx.close()


@pytest.mark.parametrize("slow_at,lens,chroma,lens_mode", [("2", "double_gauss_50mm", 0.0, 0), ("5", "petzval_58mm", 0.0, 0),
                                                           ("3", "double_gauss_50mm", 0.5, 0), ("12", "double_gauss_50mm", 0.0, 1),
                                                           ("0", "double_gauss_50mm", 0.0, 0)])
def test_stragglers_finish_in_the_cooperative_kernel(orc, monkeypatch, slow_at, lens, chroma, lens_mode):
    """Solves still running after LENTIL_SLOW_AT iterations are parked with their loop state and finished by
    solve_slow_kernel (one wave per solve, polynomial terms spread over the lanes).  With a threshold of a few
    iterations nearly every solve takes that route; the accepted-draw lists must stay bit-identical to the
    oracle's (same operations in the same order).  "0" switches the hand-over off."""
    W, H, M = 64, 48, 9
    p, model, table, keep = common.po_setup(W, H, lens=lens, samples_override=48, abb_chromatic=chroma)
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.02)
    ref = common.run_oracle(orc, p, table, visits)
    rc = ref.counters()
    monkeypatch.setenv("LENTIL_SLOW_AT", slow_at)
    monkeypatch.setenv("LENTIL_SLOW_FROM_ROUND", "0")
    monkeypatch.setenv("LENTIL_SLOW_MAX_LANES", "64")       # by default a dry wave parks only its last four lanes
    ctx = capi.Context(0)
    try:
        c = gpu_run(ctx, p, table, visits, lens_mode=lens_mode)
        assert (c.redistributed_visits, c.attempted_draws, c.accepted_draws) == (
            rc.redistributed_visits, rc.attempted_draws, rc.accepted_draws)
        if slow_at == "0":
            assert c.slow_solves == 0
        else:
            assert c.slow_solves > (1000 if int(slow_at) <= 5 else 0)      # the queue is sized for ~1 % of the solves
        check_logs(ctx, ref)
        if chroma == 0.0:
            check_frame(ctx, ref)
    finally:
        ctx.close()


@pytest.mark.parametrize("stream", ["1", "0", "outside-in"])
def test_blind_passes_and_fallback(orc, monkeypatch, stream):
    """From the second pass on nothing waits for the scan on the host: buffers are sized from what the previous pass
    found and the device checks the real counts.  Streamed (default): one scan launch publishes items and tasks to
    persistent solve waves.  LENTIL_STREAM=0: per chunk, prep_items_kernel after the chunk's scan.  Same stream
    again -> same result; a stream with many more highlights -> does not fit, the draws are redone with exact sizes."""
    W, H, M = 96, 64, 9
    p, model, table, keep = common.po_setup(W, H, samples_override=48)
    light, keep_l = common.make_stream(p, W, H, M, f_hi=0.002)      # the column arrays must outlive the passes
    heavy, keep_h = common.make_stream(p, W, H, M, f_hi=0.12, seed=0xBEEF)
    ref_l = common.run_oracle(orc, p, table, light)
    ref_h = common.run_oracle(orc, p, table, heavy)
    monkeypatch.setenv("LENTIL_CHUNKS", "3")
    if stream == "outside-in":        # the streamed pass with its scan taking tiles from both ends of the frame inwards
        monkeypatch.setenv("LENTIL_SCAN_OUTSIDE_IN", "1")
        stream = "1"
    monkeypatch.setenv("LENTIL_STREAM", stream)
    per_pass = 1 if stream == "1" else 3        # a streamed pass is one "chunk"
    ctx = capi.Context(0)
    try:
        for visits, ref, blind, fb in ((light, ref_l, 0, 0), (light, ref_l, per_pass, 0), (heavy, ref_h, per_pass, 3),
                                       (heavy, ref_h, per_pass, 0), (light, ref_l, per_pass, 0)):
            c = gpu_run(ctx, p, table, visits)
            rc = ref.counters()
            assert (c.redistributed_visits, c.attempted_draws, c.accepted_draws) == (
                rc.redistributed_visits, rc.attempted_draws, rc.accepted_draws)
            assert c.blind_chunks == blind
            assert c.streamed == (1 if stream == "1" and blind else 0)
            assert (c.fallback_chunks >= 1) if fb else (c.fallback_chunks == 0), (blind, fb, c.fallback_chunks)    # the quarter-frame chunks may just fit
            check_logs(ctx, ref)
            check_frame(ctx, ref)
    finally:
        ctx.close()


@pytest.mark.parametrize("size", ["small", "large"])
def test_streamed_pass_that_stalls_after_its_first_accept_is_run_again(orc, monkeypatch, size):
    """A streamed pass whose resident waves give up waiting (kStuckTicks) AFTER the first accept has added draws to the frame
    must not cost the frame: it is wiped and the whole pass run again in the chunked form.  LENTIL_INJECT_STALL=k makes the
    k-th streamed pass of a context stall exactly there (its first accept never closes the queue the second round's solve
    waves poll).  Passes: chunked (first of the context), streamed, streamed + stalled -> redone, streamed again -- every one
    against the oracle; the redone one reports streamed = 0, fallback_chunks = 1."""
    W, H, M, S, f_hi = (96, 64, 9, 48, 0.002) if size == "small" else (1280, 720, 9, 256, 2.0 ** -13)
    p, model, table, keep = common.po_setup(W, H, samples_override=S)
    visits, keepv = common.make_stream(p, W, H, M, f_hi=f_hi)
    ref = common.run_oracle(orc, p, table, visits) if size == "small" else common.ThreadedOracle(orc, p, table, visits, 8)
    monkeypatch.setenv("LENTIL_INJECT_STALL", "2")
    ctx = capi.Context(0)
    try:
        for k, (streamed, fb) in enumerate(((0, 0), (1, 0), (0, 1), (1, 0))):
            c = gpu_run(ctx, p, table, visits)
            rc = ref.counters()
            assert (c.streamed, c.fallback_chunks) == (streamed, fb), (k, c.streamed, c.fallback_chunks, ctx.last_redo_note())
            assert (c.redistributed_visits, c.attempted_draws, c.accepted_draws) == (
                rc.redistributed_visits, rc.attempted_draws, rc.accepted_draws)
            check_logs(ctx, ref)
            check_frame(ctx, ref)
            # why a pass was redone stays readable: who gave up waiting, and that draws had been added by then
            note = ctx.last_redo_note()
            if k < 2:
                assert note == ""
            else:
                # (the second round's resident solve wave; with LENTIL_OVERLAP_ACCEPT=1 its straggler wave may give up first)
                assert ("resident solve wave" in note or "straggler wave" in note) and "rounds_used 1" in note, note
    finally:
        ctx.close()
        ref.close()



def test_chromatic_streamed_pass_resolves_what_later_rounds_add(orc, gpu_ctx_factory):
    """abb_chromatic != 0 in a streamed pass (the second pass of a context): the frame is resolved beside the second round
    and the groups later rounds splat into are resolved again -- their accepts must flag them as such.  (Found by the
    seeded soak with another seed in round 3: accept_item_chroma flagged every round alike, the resolved image kept the
    first round's values at pixels a later round added to.)"""
    W, H, M = 59, 51, 4
    p, model, table, keep = common.po_setup(W, H, aa=2, filter_width=1.0, samples_override=33, focus_dist=400.0, abb_chromatic=0.5)
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.004, n_extra=2, seed=0xA00A)
    ref = common.run_oracle(orc, p, table, visits, n_aovs=3)
    ctx = gpu_ctx_factory()
    for again in range(3):
        c = gpu_run(ctx, p, table, visits, n_aovs=3)
        assert c.streamed == (1 if again else 0)
        check_logs(ctx, ref)
        check_frame(ctx, ref, n_aovs=3)



def test_empty_stream_and_error_paths(gpu_ctx_factory):
    p, model, table, keep = common.po_setup(32, 16)
    ctx = gpu_ctx_factory()
    with pytest.raises(capi.LentilError):
        ctx.redistribute()                      # nothing set up yet
    ctx.set_params(p)
    ctx.set_lens(table)
    ctx.alloc_frame(1)
    visits, cols = common.make_stream(p, 32, 16, 9, f_hi=0.0, v_end=0)
    ctx.upload_visits(visits)
    ctx.clear_frame()
    ctx.redistribute()
    ctx.resolve()
    buf, w = ctx.download_accum(0)
    assert not buf.any() and not w.any()
    # thin-lens abb_chromatic > 0 draws its channels from one xor128 stream: defined for one GPU, refused across GPUs
    chroma = common.tl_setup(32, 16)
    chroma.abb_chromatic = 0.5
    ctx.set_params(chroma)
    ctx.set_closest_exchange(True)
    visits, cols = common.make_stream(chroma, 32, 16, 9, f_hi=0.05)
    ctx.upload_visits(visits)
    ctx.clear_frame()
    with pytest.raises(capi.LentilError) as ei:
        ctx.redistribute()
    assert ei.value.code == _abi.ERR_UNSUPPORTED


def test_full_size_properties(gpu_ctx_factory):
    """1920x1080, 9 visits/pixel, 256 draws (BASELINE config 2): size-independent properties.
    - every visit's weight lands somewhere or is counted rejected: sum(weight) = sum over visits of
      inv_density * accepted/samples  (energy bookkeeping), checked through the counters;
    - linearity: doubling the radiance doubles the accumulators (weights unchanged);
    - the resolved image is finite and equals acc/weight."""
    import torch
    from pota_amd import workload
    W, H, M = 1920, 1080, 9
    p, model, table, keep = common.po_setup(W, H, samples_override=256)
    dev = torch.device("cuda:0")
    n = W * H * M
    cols = workload.generate(torch, 0, n, W, H, M, f_hi=2.0 ** -12, focus_dist=150.0,
                             tan_half_fov=common.tan_half_fov(p), device=dev)
    ctx = gpu_ctx_factory()
    ctx.set_params(p)
    ctx.set_lens(table)
    ctx.set_bokeh(None)
    ctx.alloc_frame(1)
    ctx.set_draw_log(0)
    visits, keepv = capi.make_visits(cols, visits_per_pixel=M, pixels_per_row=W, ptr=lambda t: t.data_ptr())
    torch.cuda.synchronize()
    ctx.bind_visits(visits, keepv)
    ctx.clear_frame(); ctx.redistribute(); ctx.resolve(); ctx.sync()
    c = ctx.counters()
    assert c.worklist_overflow == 0 and c.redistributed_visits > 0
    buf1, w1 = ctx.download_accum(0)
    direct = (n - c.redistributed_visits) * float(p.inverse_sample_density)
    splat = c.accepted_draws * float(p.inverse_sample_density) / 256.0
    assert abs(float(w1.astype(np.float64).sum()) - (direct + splat)) / (direct + splat) < 1e-5
    img = ctx.download_aov(0)
    assert np.isfinite(img).all()
    m = w1 != 0
    assert np.allclose(img[m], buf1[m] * (np.float32(1) / w1[m])[:, None], rtol=1e-6)
    # linearity in radiance (alpha stays 1)
    cols["rgba"][:, :3] *= 2.0
    torch.cuda.synchronize()
    ctx.clear_frame(); ctx.redistribute(); ctx.sync()
    buf2, w2 = ctx.download_accum(0)
    assert float(np.abs(w2.astype(np.float64) - w1).max()) <= 1e-5 * float(w1.max())
    mm = buf1[:, 0] > 0
    assert np.allclose(buf2[mm, :3], 2.0 * buf1[mm, :3], rtol=2e-5)


def _band_parity_at_full_geometry(orc, ctx, p, table, W, M, y0, rows, f_hi, n_extra=0, kinds=None, bokeh_tables=None,
                                  orc_bokeh=None):
    """A band of rows of the full-size frame (same seeded visits as the full stream), oracle vs HIP:
    pixel coordinates, bokeh radii and draw counts are those of the full frame."""
    from pota_amd import workload
    H = p.yres_without_region
    cols = workload.generate(np, y0 * W * M, (y0 + rows) * W * M, W, H, M, f_hi=f_hi, focus_dist=150.0,
                             tan_half_fov=common.tan_half_fov(p), n_extra=n_extra)
    visits, keepv = capi.make_visits(cols, visits_per_pixel=M, pixels_per_row=W, pixel_y0=y0)
    n_aovs = 1 + n_extra
    lens = orc.orc_lens_create(C.byref(table))
    ref = oracle_lib.Frame(orc, p, n_aovs=n_aovs, kinds=kinds, keep_log=True)
    ref.run(lens, orc_bokeh, visits)
    orc.orc_lens_destroy(lens)
    assert ref.counters().redistributed_visits > 20
    c = gpu_run(ctx, p, table, visits, n_aovs=n_aovs, kinds=kinds, bokeh_tables=bokeh_tables)
    assert c.redistributed_visits == ref.counters().redistributed_visits
    assert c.attempted_draws == ref.counters().attempted_draws
    assert c.accepted_draws == ref.counters().accepted_draws
    check_logs(ctx, ref)
    check_frame(ctx, ref, n_aovs=n_aovs, kinds=kinds)
    ref.close()


def _whole_frame_bookkeeping(ctx, p, W, H, M, samples, f_hi, n_extra=0):
    """Whole frame, inputs generated on the device: weight bookkeeping, finite resolve, and the same
    accepted draws whether the stream is scanned as one chunk or two."""
    import torch
    from pota_amd import workload
    dev = torch.device("cuda:0")
    n = W * H * M
    cols = workload.generate(torch, 0, n, W, H, M, f_hi=f_hi, focus_dist=150.0,
                             tan_half_fov=common.tan_half_fov(p), device=dev, n_extra=n_extra)
    visits, keepv = capi.make_visits(cols, visits_per_pixel=M, pixels_per_row=W, ptr=lambda t: t.data_ptr())
    torch.cuda.synchronize()
    ctx.bind_visits(visits, keepv)
    ctx.set_draw_log(0)
    ctx.clear_frame(); ctx.redistribute(); ctx.resolve(); ctx.sync()
    c = ctx.counters()
    assert c.worklist_overflow == 0 and c.redistributed_visits > 100
    assert c.visits == n
    w = ctx.download_accum(0)[1].astype(np.float64)
    inv = float(p.inverse_sample_density)
    # every non-redistributed visit adds inv_density to its pixel, every accepted draw inv_density/samples
    expect = (n - c.redistributed_visits) * inv + c.accepted_draws * inv / samples
    assert abs(w.sum() - expect) / expect < 1e-5
    for a in range(1 + n_extra):
        assert np.isfinite(ctx.download_aov(a)).all()
    del cols, visits, keepv
    torch.cuda.empty_cache()
    return c


def test_config3_4k_image_bokeh(orc, gpu_ctx_factory):
    """BASELINE config 3: double-gauss + image-bokeh kernel, 3840x2160, 512 draws, beauty only."""
    import os
    tex = np.load(os.path.join(common.ROOT, "tests", "golden", "example_bokeh_kernel_u8.npy")).astype(np.float32) / np.float32(255)
    tables = bokeh.build_tables(tex)
    W, H, M = 3840, 2160, 9
    p, model, table, keep = common.po_setup(W, H, samples_override=512, bokeh_enable_image=1)
    bt = _abi.BokehTable()
    bt.x, bt.y = tables["x"], tables["y"]
    for k in ("cdfRow", "rowIndices", "cdfColumn", "columnIndices"):
        setattr(bt, k, tables[k].ctypes.data)
    ob = orc.orc_bokeh_from_tables(C.byref(bt))
    ctx = gpu_ctx_factory()
    for y0 in (0, 1079, 2157):            # top edge, centre, bottom edge of the 4K frame
        _band_parity_at_full_geometry(orc, ctx, p, table, W, M, y0, 3, 2.0 ** -9, bokeh_tables=tables, orc_bokeh=ob)
    orc.orc_bokeh_destroy(ob)
    ctx.alloc_frame(1)
    _whole_frame_bookkeeping(ctx, p, W, H, M, 512, 2.0 ** -14)


def test_config4_4k_petzval_8_aovs(orc, gpu_ctx_factory):
    """BASELINE config 4: petzval polynomial, 3840x2160, 1024 draws, beauty + 8 AOVs (two of them
    closest-filtered)."""
    W, H, M = 3840, 2160, 9
    kinds = [0, 0, 1, 0, 0, 0, 1, 0, 0]
    p, model, table, keep = common.po_setup(W, H, lens="petzval_58mm", samples_override=1024)
    ctx = gpu_ctx_factory()
    for y0 in (1, 1080):
        _band_parity_at_full_geometry(orc, ctx, p, table, W, M, y0, 2, 2.0 ** -10, n_extra=8, kinds=kinds)
    ctx.alloc_frame(9, kinds)
    _whole_frame_bookkeeping(ctx, p, W, H, M, 1024, 2.0 ** -16, n_extra=8)


def test_config5_8k_rank_partition(orc, gpu_ctx_factory):
    """BASELINE config 5 as one of its 8 ranks sees it: 7680x4320 frame, rows r mod 8 == 3, 2048 draws."""
    import torch
    from pota_amd import workload
    W, H, M, G, rank = 7680, 4320, 9, 8, 3
    p, model, table, keep = common.po_setup(W, H, samples_override=2048)
    ctx = gpu_ctx_factory()
    ctx.set_params(p); ctx.set_lens(table); ctx.set_bokeh(None)
    ctx.alloc_frame(1)
    n_local = workload.frame_visit_count(W, H, M, G, rank)
    dev = torch.device("cuda:0")
    cols = workload.generate(torch, 0, n_local, W, H, M, f_hi=2.0 ** -17, focus_dist=150.0,
                             tan_half_fov=common.tan_half_fov(p), device=dev, row_stride=G, row_offset=rank)
    visits, keepv = capi.make_visits(cols, visits_per_pixel=M, pixels_per_row=W, pixel_y0=rank, pixel_row_stride=G,
                                     ptr=lambda t: t.data_ptr())
    torch.cuda.synchronize()
    ctx.bind_visits(visits, keepv)
    ctx.set_draw_log(0)
    ctx.clear_frame(); ctx.redistribute(); ctx.resolve(); ctx.sync()
    c = ctx.counters()
    assert c.worklist_overflow == 0 and c.redistributed_visits > 50 and c.visits == n_local
    buf, w = ctx.download_accum(0)
    inv = float(p.inverse_sample_density)
    expect = (n_local - c.redistributed_visits) * inv + c.accepted_draws * inv / 2048
    assert abs(w.astype(np.float64).sum() - expect) / expect < 1e-5
    # own visits only land on the rank's rows; splats land anywhere
    wimg = w.reshape(p.yres, p.xres)
    own = np.zeros(p.yres, bool); own[rank:H:G] = True
    direct_w = M * inv
    assert np.all(wimg[own][:, :W] >= direct_w * 0.5)
    assert (wimg[~own] > 0).sum() > 0 and (wimg[~own] > 0).mean() < 0.5
