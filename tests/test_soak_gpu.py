"""Seeded soaks, randomised sweeps and child-process runs of the GPU path: marker `gpu_soak`, NOT `gpu`.

The driver's round-end run (`pytest -m gpu`) has a fixed time limit, and these are the tests whose value grows with the
number of cases one lets them run, not with running them once more per round: `tools/soak.sh [seed] [cases]` runs them
(LENTIL_SOAK_SEED / LENTIL_SOAK_CASES choose the sequence; the defaults below are the short ones).  Every fixed-case
oracle comparison stays under `-m gpu` (tests/test_gpu_parity.py and friends, whose helpers these tests use).  Without a GPU
they skip.
"""
import ctypes as C
import os

import numpy as np
import pytest

import common
import oracle_lib
from pota_amd import _abi, bokeh, capi
from test_gpu_parity import _compare_with_whole, check_frame, check_logs, gpu_run


def _no_gpu():
    try:
        import torch
        return not torch.cuda.is_available()
    except Exception:      # noqa: BLE001
        return True


pytestmark = [pytest.mark.gpu_soak, pytest.mark.skipif(_no_gpu(), reason="needs a GPU (run by tools/soak.sh on a GPU box)")]


def test_randomized_tiled_bands(orc, monkeypatch):
    """Seeded soak of the tiled multi-GPU step on one device: number of ranks, band boundaries (down to bands of two
    rows, so that draws cross several bands), frame size, highlight fraction, AOV kinds, pixel lists or packed rows.
    Every band against the same rows of a whole-frame context, two passes.  Ranks are threads; the stand-in for
    torch.distributed is the per-pair-queue one of the CPU suite (steps in which only some ranks exchange anything
    stall _InProcessDist's barrier)."""
    import os
    import threading
    from pota_amd import distributed, workload
    from test_multi_gpu import _ThreadDist
    n_cases = int(os.environ.get("LENTIL_SOAK_CASES", "4"))
    rng = np.random.default_rng(int(os.environ.get("LENTIL_SOAK_SEED", "0x7D1E"), 0))
    for case in range(n_cases):
        world = int(rng.integers(2, 6))
        W, H, M = int(rng.integers(24, 90)), int(rng.integers(4 * world, 60)), 9
        cuts = sorted(rng.choice(np.arange(2, H - 1, 2), size=world - 1, replace=False).tolist())
        bounds = [0] + [int(c) for c in cuts] + [H] if rng.integers(0, 2) else None
        f_hi = float(rng.choice([0.0015, 0.01, 0.03]))
        kinds = [[0], [0, 0], [0, 1, 0]][int(rng.integers(0, 3))]
        sparse = bool(rng.integers(0, 2))
        tag = "case %d: world %d %dx%d bounds %r f_hi %g kinds %r sparse %d" % (case, world, W, H, bounds, f_hi, kinds, sparse)
        monkeypatch.setattr(distributed, "SPARSE_EXCHANGE", sparse)
        n_aovs = len(kinds)
        p, model, table, keep = common.po_setup(W, H, samples_override=int(rng.choice([16, 48])))
        visits, cols = common.make_stream(p, W, H, M, f_hi=f_hi, n_extra=n_aovs - 1)
        ctxs = []
        try:
            whole = capi.Context(0)
            ctxs.append(whole)
            gpu_run(whole, p, table, visits, n_aovs=n_aovs, kinds=kinds)
            whole.P = p
            engines, keepalive, bands = [], [], []
            for rank in range(world):
                b_lo, b_hi = distributed.band_of(rank, world, H, p.yres, bounds)
                c = workload.generate(np, b_lo * W * M, min(b_hi, H) * W * M, W, H, M, f_hi=f_hi, focus_dist=150.0,
                                      tan_half_fov=common.tan_half_fov(p), n_extra=n_aovs - 1)
                v, kv = capi.make_visits(c, visits_per_pixel=M, pixels_per_row=W, pixel_y0=b_lo)
                ctx = capi.Context(0)
                ctxs.append(ctx)
                ctx.set_params(p); ctx.set_lens(table); ctx.set_bokeh(None)
                ctx.alloc_frame(n_aovs, kinds)
                ctx.upload_visits(v)
                keepalive.append((c, v, kv))
                engines.append(distributed.HipEngine(ctx, rows=p.yres))
                bands.append((b_lo, b_hi))
            shared, errors = _ThreadDist.Shared(world), []

            def run(rank):
                try:
                    import torch
                    for _ in range(2):
                        distributed.frame_step_bands(engines[rank], _ThreadDist(shared, rank), H, p.yres, bounds)
                        engines[rank].ctx.sync()
                        torch.cuda.synchronize()
                except Exception as e:
                    errors.append((rank, e))
                    shared.barrier.abort()

            th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
            for t in th:
                t.start()
            for t in th:
                t.join(timeout=120)
            assert not errors, "%s: rank %d: %r" % (tag, errors[0][0], errors[0][1])
            assert not any(t.is_alive() for t in th), tag
            for rank in range(world):
                try:
                    _compare_with_whole(engines[rank].ctx, whole, kinds, rows=bands[rank])
                except AssertionError as e:
                    raise AssertionError("%s, rank %d: %s" % (tag, rank, e))
        finally:
            for c in ctxs:
                c.close()


@pytest.mark.parametrize("queues", ["1", "2"])
def test_passes_under_a_runtime_with_few_hardware_queues(queues):
    """GPU_MAX_HW_QUEUES=1 / 2 (read by the ROCm runtime when it initialises, hence a child process): the streams of a
    streamed pass then share hardware queues and its resident kernels can sit behind the ones they wait for.
    lentil_hip_create's probe sees that (lentil_hip_streams_concurrent = 0) and the context takes the chunked form; whatever
    form runs, four passes in a row -- first of the context, then blind ones -- must give the oracle's frame."""
    import subprocess
    import sys
    import textwrap
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r)
        import common, oracle_lib
        from pota_amd import capi
        from test_gpu_parity import check_frame, check_logs, gpu_run
        W, H, M = 320, 180, 9
        p, model, table, keep = common.po_setup(W, H, samples_override=96)
        visits, keepv = common.make_stream(p, W, H, M, f_hi=2.0 ** -11)
        orc = oracle_lib.load()
        ref = common.ThreadedOracle(orc, p, table, visits, 8)
        ctx = capi.Context(0)
        forms = []
        for k in range(4):
            c = gpu_run(ctx, p, table, visits)
            rc = ref.counters()
            assert (c.redistributed_visits, c.attempted_draws, c.accepted_draws) == (rc.redistributed_visits, rc.attempted_draws, rc.accepted_draws)
            check_logs(ctx, ref)
            check_frame(ctx, ref)
            forms.append((int(c.streamed), int(c.fallback_chunks)))
        print("FORMS", ctx.streams_concurrent(), forms)
        # a pass whose waves gave up waiting (250 ms, then redone: fallback_chunks) is the last streamed one of such a context
        stalled = [i for i, f in enumerate(forms) if f[1]]
        assert len(stalled) <= 1 and all(f == (0, 0) for f in (forms[stalled[0] + 1:] if stalled else [])), forms
        ctx.close()
        """) % (os.path.join(common.ROOT, "tests"),)
    env = dict(os.environ, GPU_MAX_HW_QUEUES=queues)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "FORMS" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    print(r.stdout.strip().splitlines()[-1])


def test_randomized_configurations_two_passes_each(orc):
    """Seeded soak over the knobs the fixed cases above hold still: frame sizes that are not multiples of the
    tile, visits per pixel, lens, compiled / table kernels, draw counts, extra AOVs, highlight fraction, focus
    distance, bokeh image, chromatic aberration.  One context per configuration, two passes (the second one is
    enqueued blind) -- counters, accepted-draw lists and frames against the oracle both times.
    LENTIL_SOAK_CASES / LENTIL_SOAK_SEED run a longer or a different sequence."""
    import os
    n_cases = int(os.environ.get("LENTIL_SOAK_CASES", "14"))      # (from case 8 on: bokeh images, chromatic aberration)
    rng = np.random.default_rng(int(os.environ.get("LENTIL_SOAK_SEED", "0x10E7"), 0))
    tex = np.load(os.path.join(common.ROOT, "tests", "golden", "example_bokeh_kernel_u8.npy")).astype(np.float32) / np.float32(255)
    tables = bokeh.build_tables(tex)
    bt = _abi.BokehTable()
    bt.x, bt.y = tables["x"], tables["y"]
    for k in ("cdfRow", "rowIndices", "cdfColumn", "columnIndices"):
        setattr(bt, k, tables[k].ctypes.data)
    for case in range(n_cases):
        W, H = int(rng.integers(17, 120)), int(rng.integers(9, 70))
        aa, fw = [(2, 1.0), (3, 1.0), (4, 1.5)][int(rng.integers(0, 3))]
        M = {2: 4, 3: 9, 4: 36}[aa]
        lens = ["double_gauss_50mm", "petzval_58mm"][int(rng.integers(0, 2))]
        override = int(rng.choice([0, 8, 17, 33]))
        n_extra = int(rng.integers(0, 4))
        f_hi = float(rng.choice([0.0005, 0.004, 0.02]))
        focus = float(rng.choice([60.0, 150.0, 400.0]))
        lens_mode = int(rng.integers(0, 2))
        image = case >= 8 and int(rng.integers(0, 4)) == 0           # the first eight cases: as first committed
        chroma = float(rng.choice([0.0, 0.0, 0.5])) if case >= 8 else 0.0
        tag = "case %d: %dx%d M=%d %s override=%d extra=%d f_hi=%g focus=%g mode=%d image=%d chroma=%g" % (
            case, W, H, M, lens, override, n_extra, f_hi, focus, lens_mode, image, chroma)
        p, model, table, keep = common.po_setup(W, H, lens=lens, aa=aa, filter_width=fw, samples_override=override,
                                                focus_dist=focus, bokeh_enable_image=int(image), abb_chromatic=chroma)
        visits, cols = common.make_stream(p, W, H, M, f_hi=f_hi, n_extra=n_extra, seed=0xA000 + case)
        ob = orc.orc_bokeh_from_tables(C.byref(bt)) if image else None
        ref = common.run_oracle(orc, p, table, visits, n_aovs=1 + n_extra, bokeh=ob)
        rc = ref.counters()
        ctx = capi.Context(0)
        try:
            for again in range(2):
                c = gpu_run(ctx, p, table, visits, n_aovs=1 + n_extra, lens_mode=lens_mode,
                            bokeh_tables=tables if image else None)
                assert (c.redistributed_visits, c.attempted_draws, c.accepted_draws) == (
                    rc.redistributed_visits, rc.attempted_draws, rc.accepted_draws), tag
                if again and rc.redistributed_visits:
                    assert c.blind_chunks > 0, tag
                check_logs(ctx, ref)
                # chromatic draws: three splats per attempt, see test_po_chromatic_aberration for the wider bound
                try:
                    check_frame(ctx, ref, n_aovs=1 + n_extra)
                except AssertionError as e:
                    raise AssertionError("%s, pass %d (streamed %d): %s" % (tag, again, int(c.streamed), e)) from e
        finally:
            ctx.close()
            if ob:
                orc.orc_bokeh_destroy(ob)


def test_randomized_thinlens_configurations(orc):
    """The thin-lens draw under the same kind of seeded soak: frame size, visits per pixel, draw count, extra AOVs,
    aperture blades, coma, optical vignetting, distortion, bokeh image; two passes per context."""
    import os
    n_cases = int(os.environ.get("LENTIL_SOAK_CASES", "8"))
    rng = np.random.default_rng(int(os.environ.get("LENTIL_SOAK_SEED", "0x71E5"), 0))
    tex = np.load(os.path.join(common.ROOT, "tests", "golden", "example_bokeh_kernel_u8.npy")).astype(np.float32) / np.float32(255)
    tables = bokeh.build_tables(tex)
    bt = _abi.BokehTable()
    bt.x, bt.y = tables["x"], tables["y"]
    for k in ("cdfRow", "rowIndices", "cdfColumn", "columnIndices"):
        setattr(bt, k, tables[k].ctypes.data)
    for case in range(n_cases):
        W, H = int(rng.integers(17, 120)), int(rng.integers(9, 70))
        aa, fw = [(2, 1.0), (3, 1.0), (4, 1.5)][int(rng.integers(0, 3))]
        M = {2: 4, 3: 9, 4: 36}[aa]
        override = int(rng.choice([0, 8, 17, 64]))
        n_extra = int(rng.integers(0, 3))
        f_hi = float(rng.choice([0.0005, 0.004, 0.02]))
        kw = {}
        if rng.integers(0, 2):
            kw["bokeh_aperture_blades"] = int(rng.choice([5, 6, 8]))
        if rng.integers(0, 2):
            kw["abb_coma"] = float(rng.choice([0.35, 1.0]))
        if rng.integers(0, 2):
            kw["optical_vignetting_distance"] = 2.0
            kw["optical_vignetting_radius"] = float(rng.choice([1.0, 1.5]))
        if rng.integers(0, 2):
            kw["abb_distortion"] = float(rng.choice([0.05, 0.15]))
        image = int(rng.integers(0, 4)) == 0
        if image:
            kw["bokeh_enable_image"] = 1
        tag = "case %d: %dx%d M=%d override=%d extra=%d f_hi=%g %r" % (case, W, H, M, override, n_extra, f_hi, kw)
        p = common.tl_setup(W, H, aa=aa, filter_width=fw, samples_override=override, **kw)
        visits, cols = common.make_stream(p, W, H, M, f_hi=f_hi, n_extra=n_extra, seed=0xB000 + case)
        ob = orc.orc_bokeh_from_tables(C.byref(bt)) if image else None
        ref = common.run_oracle(orc, p, None, visits, n_aovs=1 + n_extra, bokeh=ob)
        rc = ref.counters()
        ctx = capi.Context(0)
        try:
            for again in range(2):
                c = gpu_run(ctx, p, None, visits, n_aovs=1 + n_extra, bokeh_tables=tables if image else None)
                assert (c.redistributed_visits, c.attempted_draws, c.accepted_draws) == (
                    rc.redistributed_visits, rc.attempted_draws, rc.accepted_draws), tag
                check_logs(ctx, ref)
                # (draws of neighbouring highlights pile up on few pixels here -- no lens to spread them)
                check_frame(ctx, ref, n_aovs=1 + n_extra)
        finally:
            ctx.close()
            if ob:
                orc.orc_bokeh_destroy(ob)


def test_randomized_thinlens_chromatic(orc):
    """Thin lens with abb_chromatic > 0 (the xor128 channel stream, src/lentil_filter.cpp:393-406) under a seeded soak of
    the draw's other options: aperture blades / image, coma, optical vignetting, distortion, visits per pixel, draw
    count, extra AOVs.  Two passes per context: the second continues the generator where the first stopped."""
    import os
    n_cases = int(os.environ.get("LENTIL_SOAK_CASES", "6"))
    rng = np.random.default_rng(int(os.environ.get("LENTIL_SOAK_SEED", "0xC4A0"), 0))
    tex = np.load(os.path.join(common.ROOT, "tests", "golden", "example_bokeh_kernel_u8.npy")).astype(np.float32) / np.float32(255)
    tables = bokeh.build_tables(tex)
    bt = _abi.BokehTable()
    bt.x, bt.y = tables["x"], tables["y"]
    for k in ("cdfRow", "rowIndices", "cdfColumn", "columnIndices"):
        setattr(bt, k, tables[k].ctypes.data)
    for case in range(n_cases):
        W, H = int(rng.integers(24, 110)), int(rng.integers(16, 70))
        aa, fw = [(2, 1.0), (3, 1.0)][int(rng.integers(0, 2))]
        M = {2: 4, 3: 9}[aa]
        override = int(rng.choice([0, 8, 33]))
        n_extra = int(rng.integers(0, 2))
        f_hi = float(rng.choice([0.002, 0.02]))
        kw = {"abb_chromatic": float(rng.choice([0.3, 1.0])), "abb_chromatic_type": int(rng.integers(0, 2))}
        if rng.integers(0, 2):
            kw["bokeh_aperture_blades"] = int(rng.choice([5, 8]))
        if rng.integers(0, 2):
            kw["abb_coma"] = 0.35
        if rng.integers(0, 2):
            kw["optical_vignetting_distance"] = 2.0
            kw["optical_vignetting_radius"] = float(rng.choice([1.0, 1.5]))
        if rng.integers(0, 2):
            kw["abb_distortion"] = 0.1
        image = int(rng.integers(0, 3)) == 0
        if image:
            kw["bokeh_enable_image"] = 1
        tag = "case %d: %dx%d M=%d override=%d extra=%d f_hi=%g %r" % (case, W, H, M, override, n_extra, f_hi, kw)
        p = common.tl_setup(W, H, aa=aa, filter_width=fw, samples_override=override, **kw)
        visits, cols = common.make_stream(p, W, H, M, f_hi=f_hi, n_extra=n_extra, seed=0xC000 + case)
        ob = orc.orc_bokeh_from_tables(C.byref(bt)) if image else None
        ctx = capi.Context(0)
        state = None
        try:
            for again in range(2):
                ref = oracle_lib.Frame(orc, p, n_aovs=1 + n_extra, keep_log=True)
                if state is not None:
                    orc.orc_frame_set_xor128(ref.h, (C.c_uint32 * 4)(*state))
                ref.run(None, ob, visits)
                rc = ref.counters()
                c = gpu_run(ctx, p, None, visits, n_aovs=1 + n_extra, bokeh_tables=tables if image else None)
                assert (c.redistributed_visits, c.attempted_draws, c.accepted_draws) == (
                    rc.redistributed_visits, rc.attempted_draws, rc.accepted_draws), tag
                check_logs(ctx, ref)
                st = (C.c_uint32 * 4)()
                orc.orc_frame_get_xor128(ref.h, st)
                state = list(st)
                assert ctx.get_xor128_state() == state, tag
                check_frame(ctx, ref, n_aovs=1 + n_extra)
                ref.close()
        finally:
            ctx.close()
            if ob:
                orc.orc_bokeh_destroy(ob)
