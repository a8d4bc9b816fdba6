"""The native multi-GPU exchange of liblentil_hip.so (lentil_hip_comm_* / _allreduce / _exchange_bands,
pota_amd/csrc/lentil_comm.h) on the one GPU a test box has.

Real RCCL refuses two ranks on one device, so the world_size > 1 cases run every rank as a thread with a context of
its own and tests/fake_rccl/libfake_rccl.so (test infrastructure: an in-process stand-in for the eight RCCL entry
points the library binds, strict about the matching of sends and receives) in RCCL's place -- what is under test is
the library's side: which rows go where in which form, the metadata all-gather, the merge order, the band resolve.
The real librccl.so is exercised at world_size 1 in a process of its own (the RCCL binding is process-wide).
Every band / frame is compared with a context that processed the whole frame, which the parity suite pins to the oracle.
"""
import os
import threading

import numpy as np
import pytest

import common
from pota_amd import capi, distributed, workload
from test_gpu_parity import TOL, _compare_with_whole, gpu_run

pytestmark = pytest.mark.gpu
FAKE = os.path.join(common.ROOT, "tests", "fake_rccl", "libfake_rccl.so")


@pytest.fixture()
def fake_rccl(monkeypatch):
    if not os.path.exists(FAKE):
        pytest.fail("tests/fake_rccl/libfake_rccl.so missing: run __graft_entry__.build()")
    monkeypatch.setenv("LENTIL_RCCL_LIB", FAKE)      # read once, at the first comm call of the process


def _threads(fn, world, timeout=240):
    errors = []

    def run(rank):
        try:
            fn(rank)
        except Exception as e:          # pragma: no cover
            errors.append((rank, e))

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=timeout)
    assert not errors, "rank %d: %r" % errors[0]
    assert not any(t.is_alive() for t in th)


def _band_contexts(factory, p, table, W, H, M, world, bounds, f_hi, kinds):
    ctxs, keep, bands = [], [], []
    for rank in range(world):
        b_lo, b_hi = distributed.band_of(rank, world, H, p.yres, bounds)
        c = workload.generate(np, b_lo * W * M, min(b_hi, H) * W * M, W, H, M, f_hi=f_hi, focus_dist=150.0,
                              tan_half_fov=common.tan_half_fov(p), n_extra=len(kinds) - 1)
        v, kv = capi.make_visits(c, visits_per_pixel=M, pixels_per_row=W, pixel_y0=b_lo)
        ctx = factory()
        ctx.set_params(p); ctx.set_lens(table); ctx.set_bokeh(None)
        ctx.alloc_frame(len(kinds), kinds)
        ctx.upload_visits(v)
        keep.append((c, v, kv))
        ctxs.append(ctx)
        bands.append((b_lo, b_hi))
    return ctxs, keep, bands


@pytest.mark.parametrize("world,bounds,f_hi,sparse", [(2, None, 0.03, True), (3, None, 0.03, True), (3, [0, 9, 31, 45], 0.03, False),
                                                     (3, None, 0.0015, True), (4, [0, 2, 4, 30, 45], 0.03, True)],
                         ids=["2", "3", "3-unequal-rows", "3-sparse", "4-thin-bands"])
def test_exchange_bands_matches_the_whole_frame(orc, gpu_ctx_factory, fake_rccl, monkeypatch, world, bounds, f_hi, sparse):
    """lentil_hip_exchange_bands: every band equals the same rows of a whole-frame context, two passes (the second
    blind, with the row-limited clear).  "4-thin-bands": two-row bands, draws cross several bands and a rank exchanges
    with ranks that are not its neighbours."""
    monkeypatch.setattr(distributed, "SPARSE_EXCHANGE", sparse)
    W, H, M = 64, 45, 9
    kinds = [0, 1, 0]
    p, model, table, keep = common.po_setup(W, H, samples_override=48)
    visits, cols = common.make_stream(p, W, H, M, f_hi=f_hi, n_extra=2)
    whole = gpu_ctx_factory()
    gpu_run(whole, p, table, visits, n_aovs=3, kinds=kinds)
    whole.P = p
    ctxs, keepalive, bands = _band_contexts(gpu_ctx_factory, p, table, W, H, M, world, bounds, f_hi, kinds)
    uid = capi.Context.comm_unique_id()
    got, counts = {}, {}

    def rank_fn(rank):
        ctx = ctxs[rank]
        ctx.comm_init(uid, rank, world)
        for _ in range(2):
            got[rank] = distributed.frame_step_bands_native(ctx, H, bounds)
            ctx.sync()
        counts[rank] = ctx.exchange_counts()
        ctx.comm_destroy()

    _threads(rank_fn, world)
    reach = 0
    for rank in range(world):
        assert got[rank] == bands[rank]
        # (sparse: the fixed-capacity form, whose messages are sized before the pass -- at these frame sizes a message holds
        # every pixel of its band, nothing overflows; whole rows: the sized form)
        assert counts[rank] == ((2, 0) if sparse else (0, 0))
        lo, hi = ctxs[rank].touched_rows()
        reach = max(reach, bands[rank][0] - lo, hi - bands[rank][1])
        _compare_with_whole(ctxs[rank], whole, kinds, rows=bands[rank])
    assert reach > 0


@pytest.mark.parametrize("world,bounds", [(2, None), (4, [0, 6, 20, 33, 45])], ids=["2", "4"])
def test_exchange_bands_fixed_form_overflow_and_adaptation(orc, gpu_ctx_factory, fake_rccl, world, bounds):
    """The fixed-capacity form of lentil_hip_exchange_bands (exchange_bands_fixed, lentil_comm.h): a message's capacity is
    what both of its ends derive from the pair's previous message.  Run in a child process with LENTIL_EXCHANGE_CAP_FIRST=8 (read
    once per process), so first messages are too small for what the draws scatter over a neighbour's band: those pairs must
    detect it from the header alone (nothing merged), send the region's rows whole, and size the next message to fit --
    every exchange's band against the whole-frame context."""
    import subprocess
    import sys
    import textwrap
    code = textwrap.dedent("""
        import sys, threading, numpy as np
        sys.path.insert(0, %r)
        import common, oracle_lib
        from pota_amd import capi, distributed
        from test_gpu_parity import _compare_with_whole, gpu_run
        from test_native_exchange import _band_contexts, _threads
        world, bounds = %d, %r
        W, H, M, kinds = 64, 45, 9, [0, 1, 0]
        p, model, table, keep = common.po_setup(W, H, samples_override=48)
        visits, cols = common.make_stream(p, W, H, M, f_hi=0.03, n_extra=2)
        whole = capi.Context(0)
        gpu_run(whole, p, table, visits, n_aovs=3, kinds=kinds)
        whole.P = p
        ctxs, keepalive, bands = _band_contexts(lambda: capi.Context(0), p, table, W, H, M, world, bounds, 0.03, kinds)
        uid = capi.Context.comm_unique_id()
        counts = {}
        def rank_fn(rank):
            ctx = ctxs[rank]
            ctx.comm_init(uid, rank, world)
            counts[rank] = []
            for k in range(3):
                distributed.frame_step_bands_native(ctx, H, bounds)
                ctx.sync()
                counts[rank].append(ctx.exchange_counts())
                _compare_with_whole(ctx, whole, kinds, rows=bands[rank])
            ctx.comm_destroy()
        _threads(rank_fn, world)
        first = sum(counts[r][0][1] for r in range(world))
        later = sum(counts[r][2][1] - counts[r][0][1] for r in range(world))
        assert all(counts[r][2][0] == 3 for r in range(world)), counts
        assert first >= 2 and later == 0, counts      # overflowed at first (seen by both ends of a pair), sized to fit afterwards
        print("OVERFLOWED", first)
        """) % (os.path.join(common.ROOT, "tests"), world, bounds)
    env = dict(os.environ, LENTIL_EXCHANGE_CAP_FIRST="8", LENTIL_RCCL_LIB=FAKE)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OVERFLOWED" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_allreduce_matches_the_whole_frame(orc, gpu_ctx_factory, fake_rccl):
    """lentil_hip_allreduce: rows r mod 3 in three contexts, winner keys min-reduced + gathered, accumulators summed:
    every rank ends with the whole frame."""
    W, H, M, world = 64, 40, 9, 3
    kinds = [0, 1, 0]
    p, model, table, keep = common.po_setup(W, H, samples_override=48)
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.03, n_extra=2)
    whole = gpu_ctx_factory()
    gpu_run(whole, p, table, visits, n_aovs=3, kinds=kinds)
    whole.P = p
    ctxs, keepalive = [], []
    for rank in range(world):
        n_local = workload.frame_visit_count(W, H, M, world, rank)
        c = workload.generate(np, 0, n_local, W, H, M, f_hi=0.03, focus_dist=150.0, tan_half_fov=common.tan_half_fov(p),
                              row_stride=world, row_offset=rank, n_extra=2)
        v, kv = capi.make_visits(c, visits_per_pixel=M, pixels_per_row=W, pixel_y0=rank, pixel_row_stride=world)
        ctx = gpu_ctx_factory()
        ctx.set_params(p); ctx.set_lens(table); ctx.set_bokeh(None)
        ctx.alloc_frame(3, kinds)
        ctx.upload_visits(v)
        keepalive.append((c, v, kv))
        ctxs.append(ctx)
    uid = capi.Context.comm_unique_id()

    def rank_fn(rank):
        ctxs[rank].comm_init(uid, rank, world)
        for _ in range(2):
            distributed.frame_step_native(ctxs[rank])
            ctxs[rank].sync()

    _threads(rank_fn, world)
    for rank in range(world):
        _compare_with_whole(ctxs[rank], whole, kinds)


def test_exchange_bands_seeded_soak(orc, fake_rccl, monkeypatch):
    """Seeded soak of the native tiled step: 2-5 ranks, even or random boundaries down to two-row bands, frame size,
    highlight fraction, AOV kinds, pixel lists or packed rows; every band against the whole frame, two passes."""
    n_cases = int(os.environ.get("LENTIL_SOAK_CASES", "6"))
    rng = np.random.default_rng(int(os.environ.get("LENTIL_SOAK_SEED", "0x7D1E"), 0))
    for case in range(n_cases):
        world = int(rng.integers(2, 6))
        W, H, M = int(rng.integers(24, 90)), int(rng.integers(4 * world, 60)), 9
        cuts = sorted(rng.choice(np.arange(2, H - 1, 2), size=world - 1, replace=False).tolist())
        bounds = [0] + [int(c) for c in cuts] + [H] if rng.integers(0, 2) else None
        f_hi = float(rng.choice([0.0, 0.0015, 0.01, 0.03]))
        kinds = [[0], [0, 0], [0, 1, 0]][int(rng.integers(0, 3))]
        sparse = bool(rng.integers(0, 2))
        tag = "case %d: world %d %dx%d bounds %r f_hi %g kinds %r sparse %d" % (case, world, W, H, bounds, f_hi, kinds, sparse)
        monkeypatch.setattr(distributed, "SPARSE_EXCHANGE", sparse)
        p, model, table, keep = common.po_setup(W, H, samples_override=int(rng.choice([16, 48])))
        visits, cols = common.make_stream(p, W, H, M, f_hi=f_hi, n_extra=len(kinds) - 1)
        made = []

        def factory():
            made.append(capi.Context(0))
            return made[-1]

        try:
            whole = factory()
            gpu_run(whole, p, table, visits, n_aovs=len(kinds), kinds=kinds)
            whole.P = p
            ctxs, keepalive, bands = _band_contexts(factory, p, table, W, H, M, world, bounds, f_hi, kinds)
            uid = capi.Context.comm_unique_id()

            def rank_fn(rank):
                ctxs[rank].comm_init(uid, rank, world)
                for _ in range(2):
                    assert distributed.frame_step_bands_native(ctxs[rank], H, bounds) == bands[rank]
                    ctxs[rank].sync()

            _threads(rank_fn, world)
            for rank in range(world):
                try:
                    if f_hi > 0:
                        _compare_with_whole(ctxs[rank], whole, kinds, rows=bands[rank])
                    else:                   # nothing redistributed: direct sums only, bit for bit
                        for a in range(len(kinds)):
                            lo, hi = bands[rank][0] * p.xres, bands[rank][1] * p.xres
                            assert np.array_equal(ctxs[rank].download_aov(a)[lo:hi], whole.download_aov(a)[lo:hi])
                except AssertionError as e:
                    raise AssertionError("%s, rank %d: %s" % (tag, rank, e))
        finally:
            for c in made:
                c.close()


@pytest.mark.parametrize("partition", ["bands", "interleaved"])
def test_lentil_debug_aov_across_ranks(orc, gpu_ctx_factory, fake_rccl, partition):
    """The lentil_debug AOV (own z-buffer fed by redistributed draws only, value = the winner's draw count,
    src/lentil.h:838-845) through both exchanges: its key plane travels / is min-reduced beside the ordinary one.
    Every rank's part equals the single-context frame, which test_lentil_debug_aov pins to the oracle."""
    from pota_amd import _abi
    W, H, M, world = 64, 42, 9, 3
    kinds = [_abi.FILTER_GAUSSIAN, _abi.FILTER_CLOSEST_DEBUG, _abi.FILTER_CLOSEST]
    p, model, table, keep = common.po_setup(W, H, samples_override=0)

    def stream(v_lo, v_hi, **kw):
        c = workload.generate(np, v_lo, v_hi, W, H, M, f_hi=0.03, focus_dist=150.0, tan_half_fov=common.tan_half_fov(p),
                              n_extra=2, **kw)
        c["extra"][0] = None                   # lentil_debug has no visit column
        return c

    cw = stream(0, W * H * M)
    vw, kw_ = capi.make_visits(cw, visits_per_pixel=M, pixels_per_row=W)
    whole = gpu_ctx_factory()
    gpu_run(whole, p, table, vw, n_aovs=3, kinds=kinds)
    whole.P = p
    assert np.unique(whole.download_accum(1)[0][:, 0]).size > 3
    ctxs, keepalive, bands = [], [], []
    for rank in range(world):
        if partition == "bands":
            b_lo, b_hi = distributed.band_of(rank, world, H, p.yres)
            c = stream(b_lo * W * M, min(b_hi, H) * W * M)
            v, kv = capi.make_visits(c, visits_per_pixel=M, pixels_per_row=W, pixel_y0=b_lo)
            bands.append((b_lo, b_hi))
        else:
            c = stream(0, workload.frame_visit_count(W, H, M, world, rank), row_stride=world, row_offset=rank)
            v, kv = capi.make_visits(c, visits_per_pixel=M, pixels_per_row=W, pixel_y0=rank, pixel_row_stride=world)
            bands.append(None)
        ctx = gpu_ctx_factory()
        ctx.set_params(p); ctx.set_lens(table); ctx.set_bokeh(None)
        ctx.alloc_frame(3, kinds)
        ctx.upload_visits(v)
        keepalive.append((c, v, kv))
        ctxs.append(ctx)
    uid = capi.Context.comm_unique_id()

    def rank_fn(rank):
        ctxs[rank].comm_init(uid, rank, world)
        for _ in range(2):
            if partition == "bands":
                assert distributed.frame_step_bands_native(ctxs[rank], H) == bands[rank]
            else:
                distributed.frame_step_native(ctxs[rank])
            ctxs[rank].sync()

    _threads(rank_fn, world)
    for rank in range(world):
        _compare_with_whole(ctxs[rank], whole, [0, 1, 1], rows=bands[rank])
    # without the library's communicator the debug AOV cannot be exchanged, and says so
    lone = gpu_ctx_factory()
    lone.set_params(p); lone.set_lens(table); lone.set_bokeh(None); lone.alloc_frame(3, kinds)
    lone.upload_visits(vw)
    lone.set_closest_exchange(True)
    with pytest.raises(capi.LentilError):
        lone.redistribute()
    with pytest.raises(capi.LentilError):
        lone.compact_rows(0, 4, 0, 0, 0, 16)


def _real_rccl_world1(q):
    """world_size 1 through the real librccl.so: communicator, both exchanges (which then move nothing), the results"""
    import sys
    sys.path.insert(0, common.ROOT)
    sys.path.insert(0, os.path.join(common.ROOT, "tests"))
    os.environ.pop("LENTIL_RCCL_LIB", None)
    W, H, M = 64, 40, 9
    kinds = [0, 1, 0]
    p, model, table, keep = common.po_setup(W, H, samples_override=48)
    c = workload.generate(np, 0, W * H * M, W, H, M, f_hi=0.03, focus_dist=150.0, tan_half_fov=common.tan_half_fov(p), n_extra=2)
    v, kv = capi.make_visits(c, visits_per_pixel=M, pixels_per_row=W)
    out = {}
    ctx = capi.Context(0)
    ctx.set_params(p); ctx.set_lens(table); ctx.set_bokeh(None)
    ctx.alloc_frame(3, kinds)
    ctx.upload_visits(v)
    ctx.redistribute(); ctx.resolve(); ctx.sync()
    out["plain"] = [ctx.download_aov(a) for a in range(3)]
    ctx.comm_init(capi.Context.comm_unique_id(), 0, 1)
    distributed.frame_step_native(ctx)
    ctx.sync()
    out["allreduce"] = [ctx.download_aov(a) for a in range(3)]
    out["band"] = distributed.frame_step_bands_native(ctx, H)
    ctx.sync()
    out["bands"] = [ctx.download_aov(a) for a in range(3)]
    out["yres"] = int(p.yres)
    ctx.comm_destroy()
    ctx.close()
    q.put(out)


def test_real_rccl_world_size_one():
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    pr = mpc.Process(target=_real_rccl_world1, args=(q,))
    pr.start()
    out = q.get(timeout=300)
    pr.join(timeout=120)
    assert pr.exitcode == 0
    assert out["band"] == (0, out["yres"])
    for a, kind in enumerate([0, 1, 0]):
        ref = out["plain"][a]
        assert np.count_nonzero(ref) > 0
        for name in ("allreduce", "bands"):
            got = out[name][a]
            if kind:
                assert np.array_equal(got, ref)
            else:
                m = ref != 0
                assert np.array_equal(got != 0, m)
                assert float(np.max(np.abs(got[m] - ref[m]) / np.abs(ref[m]))) < 2 * TOL


@pytest.mark.parametrize("world,bounds", [(2, None), (3, [0, 7, 30, 45])], ids=["2", "3-unequal"])
def test_exchange_bands_carries_cryptomatte(orc, gpu_ctx_factory, fake_rccl, world, bounds):
    """Cryptomatte AOVs across GPUs: what a rank's draws add to the id maps of pixels in another rank's band travels with
    lentil_hip_exchange_bands (16-byte map-entry records) and is added to the owner's tables; every band's rows then
    hold the whole-frame oracle's maps -- id sets exactly, weights and totals to 1e-5 -- two passes."""
    import ctypes as C
    import oracle_lib
    from test_crypto import compare_tables, make_crypto_columns
    W, H, M = 64, 45, 9
    p, model, table, keep = common.po_setup(W, H, samples_override=48)
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.03)
    n_crypto, entries = 2, 3
    hashes, weights = make_crypto_columns(visits.n, W, M, n_crypto, entries)
    lens = orc.orc_lens_create(C.byref(table))
    ref = oracle_lib.Frame(orc, p, n_aovs=1, keep_log=False)
    ref.set_crypto(hashes, weights)
    ref.run(lens, None, visits)
    assert ref.counters().accepted_draws > 3000
    ctxs, keepalive, bands = _band_contexts(gpu_ctx_factory, p, table, W, H, M, world, bounds, 0.03, [0])
    for rank, ctx in enumerate(ctxs):
        v0, v1 = bands[rank][0] * W * M, min(bands[rank][1], H) * W * M
        cv, keepc = capi.make_crypto_visits([np.ascontiguousarray(h[v0:v1]) for h in hashes],
                                            [np.ascontiguousarray(w[v0:v1]) for w in weights])
        ctx.alloc_crypto(n_crypto, 32)
        ctx.upload_crypto(cv)
        keepalive.append((cv, keepc))
    uid = capi.Context.comm_unique_id()
    reach = {}

    def rank_fn(rank):
        ctx = ctxs[rank]
        ctx.comm_init(uid, rank, world)
        for _ in range(2):
            distributed.frame_step_bands_native(ctx, H, bounds)
            ctx.sync()
        lo, hi = ctx.touched_rows()
        reach[rank] = max(bands[rank][0] - lo, hi - bands[rank][1])
        ctx.comm_destroy()

    _threads(rank_fn, world)
    assert max(reach.values()) > 0            # draws did cross band boundaries
    for rank in range(world):
        b_lo, b_hi = bands[rank]
        pixels = range(b_lo * p.xres, min(b_hi, p.yres) * p.xres)
        compare_tables(ctxs[rank], ref, n_crypto, p.xres * p.yres, pixels=pixels)
    ref.close()
    orc.orc_lens_destroy(lens)
