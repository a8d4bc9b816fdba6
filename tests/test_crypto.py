"""Cryptomatte AOVs through the bidirectional pass: Camera::cryptomatte_construct_cache,
add_to_buffer_cryptomatte (src/lentil.h:781-819) and the imager's ranking (src/lentil_imager.cpp:121-161).

The oracle keeps the reference's std::map per pixel; the GPU keeps open-addressed tables filled with atomics.  Ids and
the set of ids per pixel are compared exactly; weights and totals are fp32 sums in a different order (1e-5, like the
gaussian AOVs); the ranked RGBA images exactly in the ids wherever the oracle's weights are further apart than that."""
import ctypes as C

import numpy as np
import pytest

import common
import oracle_lib
from pota_amd import bridge, capi, workload

TOL = 1e-5
UNUSED = np.array([0xFFFFFFFF], np.uint32).view(np.float32)[0]


def make_crypto_columns(n, W, M, n_crypto, entries, seed=7, palette=11):
    """Per cryptomatte AOV: [n, entries] ids and weights.  Ids follow the pixel column (objects are coherent in the
    image), one entry in four is unused, some weights are exactly 0 (the reference still creates the map entry),
    one id is -0.0 beside +0.0."""
    rng = np.random.default_rng(seed)
    pal = rng.standard_normal(palette).astype(np.float32) * 1e3
    pal[0], pal[1] = 0.0, -0.0
    pix_x = (np.arange(n) // M) % W
    hashes, weights = [], []
    for a in range(n_crypto):
        base = (pix_x // 6 + a) % palette
        ids = np.empty((n, entries), np.float32)
        for e in range(entries):                                   # distinct ids inside a visit's cache
            ids[:, e] = pal[(base + e * (1 + (rng.integers(0, 2, n)))) % palette] if e else pal[base]
        for e in range(1, entries):                                # make them distinct like a map's keys are
            clash = np.zeros(n, bool)
            for e2 in range(e):
                clash |= ids[:, e] == ids[:, e2]
            ids[clash, e] = (1e6 + 17.0 * e + a + np.arange(n)[clash] % 5).astype(np.float32)
        w = rng.random((n, entries)).astype(np.float32)
        w[rng.random((n, entries)) < 0.1] = 0.0
        unused = rng.random((n, entries)) < 0.25
        unused[:, 0] = False
        w[unused] = UNUSED
        hashes.append(np.ascontiguousarray(ids))
        weights.append(np.ascontiguousarray(w))
    return hashes, weights


def test_construct_cache_known_answers(orc):
    """cryptomatte_construct_cache, src/lentil.h:781-811, on hand-worked depth lists."""
    def cache(opacity, values):
        op = np.repeat(np.asarray(opacity, np.float32), 3)
        va = np.asarray(values, np.float32)
        ids = np.empty(16, np.float32); wts = np.empty(16, np.float32)
        n = orc.orc_crypto_construct_cache(len(values), op.ctypes.data, va.ctypes.data, ids.ctypes.data, wts.ctypes.data, 16)
        return dict(zip(ids[:n].tolist(), wts[:n].tolist())), ids[:n]
    # half-transparent A in front of opaque B: A 0.5, B 0.5, nothing left over
    d, _ = cache([0.5, 1.0], [3.0, 5.0])
    assert d == {3.0: 0.5, 5.0: 0.5}
    # only a quarter-opaque A: the remaining 0.75 goes to the last sample's id
    d, _ = cache([0.25], [3.0])
    assert d == {3.0: 1.0}
    # no depth samples at all: id 0 with the whole quota
    d, _ = cache([], [])
    assert d == {0.0: 1.0}
    # the same id at two depths is one entry; ids come out in the map's (ascending) order
    d, ids = cache([0.5, 0.5, 0.5], [7.0, -2.0, 7.0])
    assert list(ids) == [-2.0, 7.0]
    assert d[-2.0] == np.float32(0.25)
    assert d[7.0] == np.float32(np.float32(0.5) + np.float32(0.125)) + np.float32(0.125)     # 0.5 + 0.125, then the 0.125 left
    # a fully transparent sample still makes its (zero-weight) entry
    d, _ = cache([0.0, 1.0], [9.0, 4.0])
    assert d == {9.0: 0.0, 4.0: 1.0}


def test_oracle_crypto_conserves_weight(orc):
    """Every add spreads sample_weight over the cache's ids: with caches that sum to 1 a pixel's map sums to its total
    weight, and the total equals the RGBA AOV's filter weight (same adds, src/lentil.h:826 / :815)."""
    W, H, M = 48, 32, 9
    p, model, table, keep = common.po_setup(W, H, samples_override=32)
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.02)
    n = visits.n
    rng = np.random.default_rng(3)
    w = rng.random((n, 3)).astype(np.float32)
    w /= w.sum(1, keepdims=True)
    ids = np.stack([np.full(n, 1.0, np.float32), np.full(n, 2.0, np.float32), ((np.arange(n) // M) % 5 + 10).astype(np.float32)], 1)
    lens = orc.orc_lens_create(C.byref(table))
    fr = oracle_lib.Frame(orc, p, n_aovs=1)
    fr.set_crypto([ids], [w])
    fr.run(lens, None, visits)
    fw = fr.weight()
    out0, has0 = fr.crypto_rank(0, 0)
    out2, has2 = fr.crypto_rank(0, 2)
    assert np.array_equal(has0, fw != 0) and has2.any()       # (xres is W + 1: the region quirk leaves an empty column)
    for pix in np.nonzero(has0)[0][::7]:
        k, wt, tot = fr.crypto_pixel(0, pix)
        assert abs(wt.sum() - tot) <= 2e-5 * max(tot, 1.0)
        assert abs(tot - fw[pix]) <= 1e-5 * max(tot, 1.0)
        order = np.argsort(-wt, kind="stable")
        assert out0[pix, 0] == k[order[0]] and out0[pix, 2] == k[order[1]]
        assert np.isclose(out0[pix, 1] + out0[pix, 3] + (out2[pix, 1] + out2[pix, 3] if has2[pix] else 0.0), 1.0, atol=1e-4) or len(k) > 4
    fr.close()
    orc.orc_lens_destroy(lens)


def compare_tables(ctx, ref, n_crypto, np_, tol=TOL, exact_pixels=None, pixels=None):
    """exact_pixels: pixels no draw landed on -- their sums are the pixel's own visits in stream order, added by one
    lane: bit-identical to the sequential reference.  pixels: the ones to look at (default: all np_)"""
    worst = 0.0
    for a in range(n_crypto):
        ids, wts, tot = ctx.download_crypto_table(a)
        for pix in (range(np_) if pixels is None else pixels):
            rk, rw, rtot = ref.crypto_pixel(a, pix)
            used = ids[pix] != 0xFFFFFFFF
            gk = ids[pix][used].view(np.float32)
            gw = wts[pix][used]
            order = np.argsort(gk, kind="stable")
            gk, gw = gk[order], gw[order]
            assert gk.shape == rk.shape and np.array_equal(gk, rk), "pixel %d of crypto %d holds other ids" % (pix, a)
            if exact_pixels is not None and exact_pixels[pix]:
                assert np.array_equal(gw.view(np.uint32), rw.view(np.uint32)) and np.float32(tot[pix]) == np.float32(rtot), pix
            scale = max(float(rtot), 1e-30)
            if len(rw):
                worst = max(worst, float(np.max(np.abs(gw.astype(np.float64) - rw)) / scale))
            worst = max(worst, abs(float(tot[pix]) - rtot) / scale)
    assert worst < tol, worst
    return worst


def compare_ranks(ctx, ref, n_crypto, tol=TOL):
    checked = 0
    for a in range(n_crypto):
        for rank in (0, 2, 4):
            out, has = ctx.download_crypto(a, rank)
            rout, rhas = ref.crypto_rank(a, rank)
            assert np.array_equal(has, rhas)
            # ids: exact where the oracle's neighbouring weights are further apart than the summation tolerance
            for pix in np.nonzero(rhas)[0]:
                k, w, tot = ref.crypto_pixel(a, pix)
                ws = np.sort(w)[::-1] / max(tot, 1e-30)
                lo, hi = max(rank - 1, 0), min(rank + 2, len(ws) - 1)
                gaps = np.abs(np.diff(ws[lo:hi + 1]))
                if len(gaps) and gaps.min() < 4 * tol:
                    continue
                assert out[pix, 0] == rout[pix, 0] and out[pix, 2] == rout[pix, 2], (a, rank, pix)
                assert abs(out[pix, 1] - rout[pix, 1]) < 4 * tol and abs(out[pix, 3] - rout[pix, 3]) < 4 * tol
                checked += 1
    assert checked > 100
    return checked


CASES = {
    "po": dict(po=True),
    "po_owner_lanes": dict(po=True, tile="0"),
    # the tables' wipe put off to the pass, whose own-pixel kernel then writes every line whole (crypto_direct_tile_kernel<2>)
    "po_lazy_clear": dict(po=True, lazy="1"),
    "po_chromatic": dict(po=True, abb_chromatic=0.5),
    "thinlens": dict(po=False),
    "thinlens_chromatic": dict(po=False, abb_chromatic=0.6, abb_chromatic_type=0),
}


@pytest.mark.gpu
@pytest.mark.parametrize("case", list(CASES))
def test_crypto_tables_and_ranks_match_oracle(orc, gpu_ctx_factory, case, monkeypatch):
    """Two cryptomatte AOVs beside the beauty and a gaussian extra AOV, three passes of one context (the second and
    third are enqueued blind / streamed): per-pixel id sets exact, weights and totals to 1e-5, ranked images exact in
    the ids wherever the ranking is decided by more than the tolerance."""
    kw = dict(CASES[case])
    if "tile" in kw:
        monkeypatch.setenv("LENTIL_CRYPTO_TILE", kw.pop("tile"))      # the untiled owner-lane kernel
    if "lazy" in kw:
        monkeypatch.setenv("LENTIL_CRYPTO_LAZY_CLEAR", kw.pop("lazy"))
    W, H, M = 64, 48, 9
    if kw.pop("po"):
        p, model, table, keep = common.po_setup(W, H, samples_override=48, **kw)
    else:
        p, table = common.tl_setup(W, H, samples_override=48, **kw), None
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.02, n_extra=1)
    n = visits.n
    n_crypto, entries = 2, 3
    hashes, weights = make_crypto_columns(n, W, M, n_crypto, entries)
    cv, keepc = capi.make_crypto_visits(hashes, weights)
    lens = orc.orc_lens_create(C.byref(table)) if table is not None else None
    ctx = gpu_ctx_factory()
    import os
    if os.environ.get("LENTIL_EXPECT_SPARE_STREAM") == "1":      # (test_crypto_replay_beside_the_draws_...)
        assert ctx.streams_concurrent() == 2
    ctx.set_params(p)
    if table is not None:
        ctx.set_lens(table)
    ctx.alloc_frame(2)
    ctx.alloc_crypto(n_crypto, 32)
    ctx.upload_visits(visits)
    ctx.upload_crypto(cv)
    for frame in range(3):
        ref = oracle_lib.Frame(orc, p, n_aovs=2, keep_log=False)
        ref.set_crypto(hashes, weights)
        if case == "thinlens_chromatic":
            st = ctx.get_xor128_state()
            orc.orc_frame_set_xor128(ref.h, (C.c_uint32 * 4)(*st))
        ref.run(lens, None, visits)
        rc = ref.counters()
        assert rc.redistributed_visits > 100 and rc.accepted_draws > 3000
        ctx.clear_frame()
        ctx.redistribute()
        ctx.resolve()
        c = ctx.counters()
        assert (c.redistributed_visits, c.accepted_draws) == (rc.redistributed_visits, rc.accepted_draws)
        untouched = np.ones(p.xres * p.yres, bool)
        untouched[ctx.draw_log()[:, 2]] = False
        assert untouched.sum() > 500
        compare_tables(ctx, ref, n_crypto, p.xres * p.yres, exact_pixels=untouched)
        compare_ranks(ctx, ref, n_crypto)
        # the beauty is what it is without cryptomatte
        buf, w = ctx.download_accum(0)
        assert common.rel_err(w[ref.weight() != 0], ref.weight()[ref.weight() != 0]) < TOL
        ref.close()
    if lens:
        orc.orc_lens_destroy(lens)


@pytest.mark.gpu
def test_crypto_error_paths(orc, gpu_ctx_factory):
    """A table or draw log that is too small is an error of the pass, never a silently short map; the library sizes its
    own log for the next attempt; columns of another stream are refused."""
    W, H, M = 48, 32, 9
    p, model, table, keep = common.po_setup(W, H, samples_override=32)
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.03)
    n = visits.n
    hashes, weights = make_crypto_columns(n, W, M, 1, 4)
    cv, keepc = capi.make_crypto_visits(hashes, weights)
    ctx = gpu_ctx_factory()
    ctx.set_params(p); ctx.set_lens(table); ctx.alloc_frame(1)
    with pytest.raises(capi.LentilError):                    # nothing allocated
        ctx.upload_crypto(cv)
    ctx.alloc_crypto(1, 2)
    ctx.upload_visits(visits)
    ctx.clear_frame()
    with pytest.raises(capi.LentilError, match="columns"):   # no columns for this stream
        ctx.redistribute()
    ctx.upload_crypto(cv)
    ctx.clear_frame()
    with pytest.raises(capi.LentilError, match="table full") as e:
        ctx.redistribute()
    assert e.value.code == -4 or "full" in str(e.value)
    ctx.alloc_crypto(1, 0)                                   # default: 16 ids per pixel
    ctx.upload_crypto(cv)
    ctx.set_draw_log(64)
    ctx.clear_frame()
    with pytest.raises(capi.LentilError, match="draw log"):
        ctx.redistribute()
    ctx.set_draw_log(0)                                      # the library's own log from here on
    ctx.clear_frame()
    ctx.redistribute()
    lens = orc.orc_lens_create(C.byref(table))
    ref = oracle_lib.Frame(orc, p, n_aovs=1)
    ref.set_crypto(hashes, weights)
    ref.run(lens, None, visits)
    compare_tables(ctx, ref, 1, p.xres * p.yres)
    # between GPUs the tables travel with the tiled exchange only (tests/test_native_exchange.py): a context set up for
    # the interleaved partition's deferred closest merge refuses
    ctx.set_closest_exchange(1, 0)
    ctx.clear_frame()
    with pytest.raises(capi.LentilError, match="tiled exchange only"):
        ctx.redistribute()
    ctx.set_closest_exchange(0, 0)
    # the same visits bound anew: caches handed over for an earlier stream do not count, whatever its length
    ctx.upload_visits(visits)
    ctx.clear_frame()
    with pytest.raises(capi.LentilError, match="columns"):
        ctx.redistribute()
    ctx.upload_crypto(cv)
    ctx.clear_frame()
    ctx.redistribute()
    # a shorter stream with the old columns still bound
    visits2, cols2 = common.make_stream(p, W, H, M, f_hi=0.03, v_end=n - M * W)
    ctx.upload_visits(visits2)
    ctx.clear_frame()
    with pytest.raises(capi.LentilError, match="columns"):
        ctx.redistribute()
    ref.close()
    orc.orc_lens_destroy(lens)


@pytest.mark.gpu
def test_crypto_log_bookkeeping_survives_a_camera_update(orc, gpu_ctx_factory, monkeypatch):
    """lentil.so calls lentil_hip_alloc_frame on every camera update, which drops the cryptomatte state; the draw log the
    replay allocated itself stays.  A later frame with more accepted draws than that log holds must still find the log
    marked as the library's own: the pass reports NOMEM once, sizes the log, and the retry (what the bridge does) fits."""
    monkeypatch.setenv("LENTIL_CRYPTO_LOG", "2000")
    W, H, M = 48, 32, 9
    p, model, table, keep = common.po_setup(W, H, samples_override=32)
    ctx = gpu_ctx_factory()
    ctx.set_params(p); ctx.set_lens(table)
    lens = orc.orc_lens_create(C.byref(table))
    for frame, f_hi in enumerate((0.004, 0.03)):              # frame 2 accepts ~8x the draws of frame 1
        visits, cols = common.make_stream(p, W, H, M, f_hi=f_hi, seed=0x5EED + frame)
        hashes, weights = make_crypto_columns(visits.n, W, M, 1, 4)
        cv, keepc = capi.make_crypto_visits(hashes, weights)
        ctx.alloc_frame(1)                                    # the camera update: drops the tables and their state
        ctx.alloc_crypto(1, 0)
        ctx.upload_visits(visits)
        ctx.upload_crypto(cv)
        ctx.clear_frame()
        try:
            ctx.redistribute()
            retried = False
        except capi.LentilError as e:
            assert "draw log" in str(e)
            ctx.clear_frame()
            ctx.redistribute()                                # the library's log now fits
            retried = True
        assert retried == (frame == 1)
        ref = oracle_lib.Frame(orc, p, n_aovs=1)
        ref.set_crypto(hashes, weights)
        ref.run(lens, None, visits)
        assert ref.counters().accepted_draws > (2000 if frame else 100)
        compare_tables(ctx, ref, 1, p.xres * p.yres)
        ref.close()
    orc.orc_lens_destroy(lens)


# ----------------------------------------------------------------------------------------------------------------
# the plugin side: liblentil_bridge.so
# ----------------------------------------------------------------------------------------------------------------
def depth_lists(rng, n, max_depth=4, palette=9):
    """n AOV samples' depth lists: (opacity rgb [d, 3], id [d]) with 0..max_depth depths, ids from a small palette"""
    pal = (rng.standard_normal(palette) * 100).astype(np.float32)
    out = []
    for _ in range(n):
        d = int(rng.integers(0, max_depth + 1))
        op = rng.random((d, 3)).astype(np.float32)
        op[rng.random(d) < 0.3] = 1.0
        out.append((op, pal[rng.integers(0, palette, d)].astype(np.float32)))
    return out


def test_bridge_construct_cache_equals_oracle(orc):
    """lentil_crypto_construct_cache (what lentil_filter's filter_pixel calls per cryptomatte AOV) against the oracle's
    restatement of src/lentil.h:781-811: the same pairs, bit for bit, padded with unused pairs."""
    lib = bridge.load()
    rng = np.random.default_rng(11)
    cap = 6
    for op, ids in depth_lists(rng, 300):
        a_i = np.empty(cap, np.float32); a_w = np.empty(cap, np.float32)
        b_i = np.empty(cap, np.float32); b_w = np.empty(cap, np.float32)
        na = lib.lentil_crypto_construct_cache(len(ids), op.ctypes.data, ids.ctypes.data, a_i.ctypes.data, a_w.ctypes.data, cap)
        nb = orc.orc_crypto_construct_cache(len(ids), op.ctypes.data, ids.ctypes.data, b_i.ctypes.data, b_w.ctypes.data, cap)
        assert na == nb and na >= 1
        assert np.array_equal(a_i[:na].view(np.uint32), b_i[:nb].view(np.uint32))
        assert np.array_equal(a_w[:na].view(np.uint32), b_w[:nb].view(np.uint32))
        assert (a_w[na:].view(np.uint32) == 0xFFFFFFFF).all()
    one = np.ones(3, np.float32); v = np.arange(3, dtype=np.float32)
    small = np.empty(2, np.float32)
    assert lib.lentil_crypto_construct_cache(3, np.tile(one * 0.5, 3).ctypes.data, v.ctypes.data, small.ctypes.data,
                                             small.copy().ctypes.data, 2) == -1          # three ids do not fit two pairs


def test_crypto_aov_list_and_rank_names():
    """Camera::setup_crypto_aovs (src/lentil.h:1015-1055) and the imager's rank names (src/lentil_imager.cpp:124-126)."""
    lib = bridge.load()
    outputs = ["RGBA RGBA lentil_replaced_filter driver",
               "crypto_material RGBA crypto_filter driver",
               "crypto_material00 FLOAT crypto_filter00 driver",
               "crypto_material01 FLOAT crypto_filter01 driver",
               "cam crypto_object02 FLOAT crypto_filter02 driver HALF",
               "Z FLOAT closest driver"]
    arr = (C.c_char_p * len(outputs))(*[o.encode() for o in outputs])
    plans = (bridge.AovPlan * 8)()
    m = lib.lentil_setup_crypto_aovs(arr, len(outputs), plans, 8)
    assert m == 4
    got = [(plans[k].to.aov_name.decode(), plans[k].is_crypto, plans[k].to.filter.decode()) for k in range(m)]
    assert got == [("crypto_material", 0, "crypto_filter"), ("crypto_material00", 1, "lentil_replaced_filter"),
                   ("crypto_material01", 1, "lentil_replaced_filter"), ("crypto_object02", 1, "lentil_replaced_filter")]
    assert bridge.rebuild_output(plans[3].to) == "cam crypto_object02 FLOAT lentil_replaced_filter driver HALF"
    # sanitize keeps what lentil filters: the display AOV goes
    k = lib.lentil_sanitize_aov_list(plans, m)
    assert [plans[j].to.aov_name.decode() for j in range(k)] == ["crypto_material00", "crypto_material01", "crypto_object02"]
    assert lib.lentil_setup_crypto_aovs(arr, len(outputs), plans, 2) == -1
    ranks = {n: lib.lentil_crypto_rank_of_name(n.encode()) for n in
             ("crypto_material00", "crypto_material01", "crypto_material02", "crypto_asset01", "crypto_object02", "crypto_object03",
              "crypto_custom01")}
    assert ranks == {"crypto_material00": 0, "crypto_material01": 2, "crypto_material02": 4, "crypto_asset01": 2,
                     "crypto_object02": 4, "crypto_object03": 0, "crypto_custom01": 0}


def test_stage_keeps_the_caches_beside_their_visits():
    """lentil_stage_set_crypto / lentil_stage_crypto (host logic): appended from two thread slots, the concatenated
    cryptomatte columns line up with the concatenated visits; misuse is refused."""
    lib = bridge.load()
    W, H, M = 12, 8, 3
    p, model, table, keep = common.po_setup(W, H)
    n = W * H * M
    cols = workload.generate(np, 0, n, W, H, M, f_hi=0.05, focus_dist=150.0, tan_half_fov=common.tan_half_fov(p))
    pix = np.arange(n, dtype=np.uint32) // M
    cols["pixel"] = ((pix % W) | ((pix // W) << 16)).astype(np.uint32)
    cols["inv_density"] = np.full(n, p.inverse_sample_density, np.float32)
    rng = np.random.default_rng(1)
    ids = [rng.standard_normal((n, 2)).astype(np.float32) for _ in range(2)]
    wts = [rng.random((n, 2)).astype(np.float32) for _ in range(2)]
    cols["crypto_ids"], cols["crypto_weights"] = ids, wts
    stage = C.c_void_p()
    assert lib.lentil_stage_create(2, 0, C.byref(stage)) == 0
    cv = _abi_crypto()
    assert lib.lentil_stage_crypto(stage, C.byref(cv)) != 0                  # no cryptomatte AOVs announced
    assert lib.lentil_stage_set_crypto(stage, 2, 65) != 0                    # entries out of range
    assert lib.lentil_stage_set_crypto(stage, 2, 2) == 0
    order = [np.arange(100, n), np.arange(0, 100)]                           # slot 0 holds the later visits
    for k in range(2):
        bridge.stage_append_arrays(stage, k, cols, order[k])
    assert lib.lentil_stage_set_crypto(stage, 1, 2) != 0                     # not on a stage that holds visits
    from pota_amd import _abi
    v = _abi.Visits()
    assert lib.lentil_stage_visits(stage, C.byref(v)) == 0 and v.n == n
    assert lib.lentil_stage_crypto(stage, C.byref(cv)) == 0
    assert (cv.n, cv.n_crypto, cv.entries) == (n, 2, 2)
    perm = np.concatenate(order)
    got_rgba = np.ctypeslib.as_array(C.cast(v.rgba, C.POINTER(C.c_float)), (n, 4))
    assert np.array_equal(got_rgba, cols["rgba"][perm])
    for a in range(2):
        h = np.ctypeslib.as_array(C.cast(cv.hash[a], C.POINTER(C.c_float)), (n, 2))
        w = np.ctypeslib.as_array(C.cast(cv.weight[a], C.POINTER(C.c_float)), (n, 2))
        assert np.array_equal(h, ids[a][perm]) and np.array_equal(w, wts[a][perm])
    # a capture without caches on such a stage is refused
    c = bridge.SampleCapture()
    assert lib.lentil_stage_append(stage, 0, C.byref(c)) != 0
    lib.lentil_stage_destroy(stage)


def _abi_crypto():
    from pota_amd import _abi
    return _abi.CryptoVisits()


@pytest.mark.gpu
@pytest.mark.parametrize("streaming", [False, True], ids=["staged", "streamed-upload"])
def test_imager_crypto_buckets_match_oracle(orc, gpu_ctx_factory, monkeypatch, streaming):
    """filter_pixel's capture with cryptomatte caches (built from depth lists by lentil_crypto_construct_cache) through
    the stage and the once-only imager pass; cryptomatte buckets against the oracle's ranking, including the bucket
    rows the reference abandons at the first pixel whose map is too short.  The library sizes its draw log itself (the
    first pass of the context reports it short; the imager repeats it)."""
    lib = bridge.load()
    W, H, M = 64, 40, 9
    p, model, table, keep = common.po_setup(W, H, samples_override=48)
    n = W * H * M
    cols = workload.generate(np, 0, n, W, H, M, f_hi=0.03, focus_dist=150.0, tan_half_fov=common.tan_half_fov(p))
    pix = np.arange(n, dtype=np.uint32) // M
    cols["pixel"] = ((pix % W) | ((pix // W) << 16)).astype(np.uint32)
    cols["inv_density"] = np.full(n, p.inverse_sample_density, np.float32)
    rng = np.random.default_rng(5)
    names = ["crypto_object00", "crypto_object01", "crypto_material02"]
    entries = 4
    ids = [np.empty((n, entries), np.float32) for _ in names]
    wts = [np.empty((n, entries), np.float32) for _ in names]
    for a in range(len(names)):
        # objects coherent over 8-pixel columns: the palette of a sample follows its pixel
        lists = depth_lists(rng, n, max_depth=entries, palette=3 if a == 1 else 7)
        for v, (op, val) in enumerate(lists):
            val = val + np.float32((v // M % W) // 8 * 1000.0) if a != 1 else val
            val = np.ascontiguousarray(val, np.float32)
            got = lib.lentil_crypto_construct_cache(len(val), op.ctypes.data, val.ctypes.data, ids[a][v].ctypes.data,
                                                    wts[a][v].ctypes.data, entries)
            assert got >= 1
    cols["crypto_ids"], cols["crypto_weights"] = ids, wts
    visits, kv = capi.make_visits(cols, visits_per_pixel=0)

    lens = orc.orc_lens_create(C.byref(table))
    ref = oracle_lib.Frame(orc, p, n_aovs=1)
    ref.set_crypto(ids, wts)
    ref.run(lens, None, visits)

    ctx = gpu_ctx_factory()
    ctx.set_params(p); ctx.set_lens(table); ctx.set_bokeh(None); ctx.alloc_frame(1)
    ctx.alloc_crypto(len(names), 32)
    ctx.set_draw_log(0)
    monkeypatch.setenv("LENTIL_CRYPTO_LOG", "1000")     # the library's first guess at the draw log: too short for this frame
    stage = C.c_void_p()
    assert lib.lentil_stage_create(2, 0, C.byref(stage)) == 0
    assert lib.lentil_stage_set_crypto(stage, len(names), entries) == 0
    if streaming:      # blocks of 700 visits (and their caches) leave for the GPU while the appends go on
        assert lib.lentil_stage_stream_to(stage, ctx.h, 700, 0) == 0
    im = C.c_void_p()
    assert lib.lentil_imager_create(ctx.h, stage, C.byref(p), 1, C.byref(im)) == 0
    ranks = (C.c_int * len(names))(*[lib.lentil_crypto_rank_of_name(nm.encode()) for nm in names])
    assert list(ranks) == [0, 2, 4]
    assert lib.lentil_imager_set_crypto(im, len(names), ranks) == 0
    half = n // 2 // M * M
    bridge.stage_append_arrays(stage, 0, cols, np.arange(0, half))
    bridge.stage_append_arrays(stage, 1, cols, np.arange(half, n))
    B = 16
    for a in range(len(names)):
        rout, rhas = ref.crypto_rank(a, ranks[a])
        rout = rout.reshape(p.yres, p.xres, 4); rhas = rhas.reshape(p.yres, p.xres)
        checked = abandoned = 0
        for y in range(0, p.yres, B):
            for x in range(0, p.xres, B):
                sx, sy = min(B, p.xres - x), min(B, p.yres - y)
                buf = np.full((sy, sx, 4), -7.0, np.float32)
                rc = lib.lentil_imager_process_crypto_bucket(im, a, x, y, sx, sy, buf.ctypes.data)
                assert rc == 0, lib.lentil_imager_last_error(im)
                for j in range(sy):
                    stop = sx
                    miss = np.nonzero(~rhas[y + j, x:x + sx])[0]
                    if len(miss):
                        stop = int(miss[0]); abandoned += 1
                    assert (buf[j, stop:] == -7.0).all()              # left as it was (src/lentil_imager.cpp:132-134)
                    for i in range(stop):
                        k, w, tot = ref.crypto_pixel(a, (y + j) * p.xres + x + i)
                        ws = np.sort(w)[::-1] / max(tot, 1e-30)
                        lo, hi = max(ranks[a] - 1, 0), min(ranks[a] + 2, len(ws) - 1)
                        gaps = np.abs(np.diff(ws[lo:hi + 1]))
                        if len(gaps) and gaps.min() < 4 * TOL:
                            continue
                        assert buf[j, i, 0] == rout[y + j, x + i, 0] and buf[j, i, 2] == rout[y + j, x + i, 2]
                        assert abs(buf[j, i, 1] - rout[y + j, x + i, 1]) < 4 * TOL
                        checked += 1
        assert checked > 200
        if ranks[a]:
            assert abandoned > 0
    # the beauty went through the same pass
    buf = np.empty((p.yres, p.xres, 4), np.float32)
    assert lib.lentil_imager_process_bucket(im, 0, 0, 0, p.xres, p.yres, buf.ctypes.data) == 0
    want = ref.resolve(0).reshape(p.yres, p.xres, 4)
    assert common.rel_err(buf[want != 0], want[want != 0]) < 1e-4
    lib.lentil_imager_destroy(im)
    lib.lentil_stage_destroy(stage)
    ref.close()
    orc.orc_lens_destroy(lens)


@pytest.mark.gpu
def test_piecewise_upload_carries_the_caches(orc, gpu_ctx_factory):
    """lentil_hip_visits_begin_crypto / _append_crypto: the stream handed over in parts of uneven size, caches with every
    part (the columns grow on the way: capacity hint far too small); a second frame through the same columns; the
    tables equal the ones of a whole upload.  Misuse is refused."""
    W, H, M = 48, 32, 9
    p, model, table, keep = common.po_setup(W, H, samples_override=32)
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.03)
    n = visits.n
    hashes, weights = make_crypto_columns(n, W, M, 2, 3)
    lens = orc.orc_lens_create(C.byref(table))
    ref = oracle_lib.Frame(orc, p, n_aovs=1)
    ref.set_crypto(hashes, weights)
    ref.run(lens, None, visits)
    ctx = gpu_ctx_factory()
    ctx.set_params(p); ctx.set_lens(table); ctx.alloc_frame(1)
    layout = _abi_visits_layout(M, W)
    ctx.visits_begin(layout, 1000)
    with pytest.raises(capi.LentilError):                 # no cryptomatte AOVs allocated yet
        ctx.visits_begin_crypto(3)
    ctx.alloc_crypto(2, 32)
    NAMES = ("rgba", "pos_z", "raydir_time", "volume_ignore", "transmission")
    for frame in range(2):
        ctx.visits_begin(layout, 1000)
        ctx.visits_begin_crypto(3)
        cuts = [0, 5 * M, 500 * M, n // 2 // M * M, n - 7 * M, n] if frame == 0 else [0, n // 2 // M * M, n]
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            part = type(layout)()
            part.n = hi - lo
            for name in NAMES:
                setattr(part, name, cols[name][lo:hi].ctypes.data)
            cv, keepc = capi.make_crypto_visits([h[lo:hi] for h in hashes], [w[lo:hi] for w in weights])
            if frame == 0 and lo == 0:
                with pytest.raises(capi.LentilError, match="carries cryptomatte"):
                    ctx.visits_append(part)
            t = ctx.visits_append_crypto(part, cv)
            ctx.visits_wait(t)                             # (the slices are pageable temporaries)
        assert ctx.visits_end() == n
        ctx.clear_frame(); ctx.redistribute()
        compare_tables(ctx, ref, 2, p.xres * p.yres)
    # a plain stream after it: its appends take no caches, and the pass then misses them
    ctx.visits_begin(layout, n)
    part = type(layout)(); part.n = n
    for name in NAMES:
        setattr(part, name, cols[name].ctypes.data)
    cv, keepc = capi.make_crypto_visits(hashes, weights)
    with pytest.raises(capi.LentilError, match="without cryptomatte"):
        ctx.visits_append_crypto(part, cv)
    ctx.visits_append(part)
    ctx.visits_end()
    ctx.upload_crypto(cv)                                  # ... unless they are uploaded whole
    ctx.clear_frame(); ctx.redistribute()
    compare_tables(ctx, ref, 2, p.xres * p.yres)
    ref.close()
    orc.orc_lens_destroy(lens)


def _abi_visits_layout(M, W):
    from pota_amd import _abi
    lay = _abi.Visits()
    lay.visits_per_pixel, lay.pixels_per_row, lay.pixel_row_stride = M, W, 1
    return lay


@pytest.mark.gpu
def test_crypto_replay_beside_the_draws_in_a_process_with_a_fifth_queue():
    """With GPU_MAX_HW_QUEUES > 4 (read by the runtime when it initialises: a process of its own) a context finds a spare
    stream beside the pass's four, and a streamed pass replays the own-pixel cryptomatte adds there while its draws go
    on: the polynomial-optics case of test_crypto_tables_and_ranks_match_oracle, in such a process."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8", LENTIL_EXPECT_SPARE_STREAM="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k",
                        "test_crypto_tables_and_ranks_match_oracle and po and not owner and not chromatic and not lazy"], env=env, cwd=common.ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "1 passed" in r.stdout, r.stdout[-2000:]
