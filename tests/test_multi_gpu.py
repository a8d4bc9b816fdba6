"""N>1 path on CPU: world_size-2 gloo run of pota_amd.distributed.frame_step with a CPU engine
(the oracle does the per-rank compute -- test infrastructure), checking that
  rows r mod G partition + sum all-reduce + local resolve  ==  the single-rank result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import common


class OracleEngine:
    """CPU stand-in for HipEngine with the same interface (clear/redistribute/accum/resolve, the
    closest-AOV key exchange, and the row views / merge of the tiled mode).  Accumulators are kept as
    pixel records [np][4 * n_aovs + 1] like the HIP library's, so that rows are contiguous."""

    def __init__(self, lib, p, table, visits, cols=None, kinds=None):
        import ctypes as C
        import oracle_lib
        self.lib, self.p, self.visits, self.cols = lib, p, visits, cols
        self.kinds = list(kinds) if kinds else [0]
        self.n_aovs = len(self.kinds)
        self.lens = lib.orc_lens_create(C.byref(table))
        self.frame = None
        self.oracle_lib = oracle_lib
        self.np = p.xres * p.yres
        self.rows = p.yres
        self.stride = 4 * self.n_aovs + 1
        self.accum = torch.zeros(self.np * self.stride, dtype=torch.float32)
        self.rec = self.accum.view(self.np, self.stride)
        self.zkey = torch.zeros(self.np, dtype=torch.int64) if any(self.kinds) else None
        self.device = torch.device("cpu")
        self.deferred = False
        self.resolved = None

    def _aov(self, a):
        return self.rec[:, 4 * a:4 * a + 4]

    def _gid(self, v):
        """frame-wide visit id, same rule as visit_gid() in pota_amd/csrc/lentil_kernels.h"""
        V = self.visits
        row_visits = V.pixels_per_row * V.visits_per_pixel
        ly = v // row_visits
        return (V.pixel_y0 + ly * V.pixel_row_stride) * row_visits + (v - ly * row_visits)

    def set_deferred_closest(self, on):
        self.deferred = bool(on)

    def clear(self):
        self.frame = self.oracle_lib.Frame(self.lib, self.p, n_aovs=self.n_aovs, kinds=self.kinds, shadow=False)
        self.accum.zero_()

    def redistribute(self):
        self.frame.run(self.lens, None, self.visits)
        for a in range(self.n_aovs):
            self._aov(a)[:] = torch.from_numpy(self.frame.buffer(a))
        self.rec[:, 4 * self.n_aovs] = torch.from_numpy(self.frame.weight())
        if self.zkey is not None:
            zv = self.frame.zvisit().astype(np.int64)
            have = zv != 0xFFFFFFFF
            depth_bits = self.frame.zbuffer().view(np.uint32).astype(np.uint64)
            key = np.full(self.np, 0xFFFFFFFFFFFFFFFF, np.uint64)
            gid = self._gid(zv[have]).astype(np.uint64)
            key[have] = (depth_bits[have] << np.uint64(32)) | (np.uint64(0xFFFFFFFF) - gid)
            self.zkey[:] = torch.from_numpy(key.view(np.int64))
            if self.deferred:
                for a in range(self.n_aovs):
                    if self.kinds[a]:
                        self._aov(a).zero_()

    def finish_local(self):
        pass

    def after_key_exchange(self):
        pass

    def closest_gather(self):
        V = self.visits
        key = self.zkey.numpy().view(np.uint64)
        have = key != np.uint64(0xFFFFFFFFFFFFFFFF)
        gid = (np.uint64(0xFFFFFFFF) - (key & np.uint64(0xFFFFFFFF))).astype(np.int64)
        row_visits = V.pixels_per_row * V.visits_per_pixel
        py = gid // row_visits
        mine = have & (py >= V.pixel_y0) & ((py - V.pixel_y0) % V.pixel_row_stride == 0)
        v = ((py - V.pixel_y0) // V.pixel_row_stride) * row_visits + (gid - py * row_visits)
        mine &= v < V.n
        for a in range(1, self.n_aovs):
            if self.kinds[a]:
                self._aov(a)[torch.from_numpy(mine)] = torch.from_numpy(self.cols["extra"][a - 1][v[mine]])

    def before_resolve(self):
        pass

    # ---- tiled mode
    def touched_rows(self):
        w = self.rec.abs().sum(dim=1).view(self.p.yres, self.p.xres).sum(dim=1).numpy()
        rows = np.nonzero(w)[0]
        return (int(rows[0]), int(rows[-1]) + 1) if len(rows) else (0, 0)

    def acc_rows(self, lo, hi):
        per = self.p.xres * self.stride
        return self.accum[lo * per:hi * per]

    def key_rows(self, lo, hi):
        return self.zkey[lo * self.p.xres:hi * self.p.xres]

    def merge_rows(self, lo, acc, keys):
        n = acc.numel() // (self.p.xres * self.stride)
        p0, p1 = lo * self.p.xres, (lo + n) * self.p.xres
        src = acc.view(-1, self.stride)
        dst = self.rec[p0:p1]
        closer = None
        if keys is not None:
            mine = self.zkey[p0:p1].numpy().view(np.uint64)
            theirs = keys.numpy().view(np.uint64)
            closer = torch.from_numpy(theirs < mine)
        for a in range(self.n_aovs):
            if self.kinds[a]:
                dst[:, 4 * a:4 * a + 4][closer] = src[:, 4 * a:4 * a + 4][closer]
            else:
                dst[:, 4 * a:4 * a + 4] += src[:, 4 * a:4 * a + 4]
        dst[:, 4 * self.n_aovs] += src[:, 4 * self.n_aovs]
        if keys is not None:
            self.zkey[p0:p1] = torch.from_numpy(np.minimum(mine, theirs).view(np.int64))

    def resolve_rows(self, lo, hi):
        self.resolve()
        keep = np.zeros(self.np, bool)
        keep[lo * self.p.xres:hi * self.p.xres] = True
        for r in self.resolved:
            r[~keep] = 0

    def resolve(self):
        w = self.rec[:, 4 * self.n_aovs].numpy()
        self.resolved = []
        for a in range(self.n_aovs):
            acc = self._aov(a).numpy()
            out = acc.copy()
            if not self.kinds[a]:
                m = w != 0
                out[m] = acc[m] * (np.float32(1.0) / w[m])[:, None]
            self.resolved.append(out)


def _worker(rank, world, port, W, H, M, f_hi, samples, kinds, q, bands=False, bounds=None):
    sys.path.insert(0, common.ROOT)
    sys.path.insert(0, os.path.join(common.ROOT, "tests"))
    import oracle_lib
    from pota_amd import capi, distributed, workload
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = oracle_lib.load()
    p, model, table, keep = common.po_setup(W, H, samples_override=samples)
    if bands:
        b_lo, b_hi = distributed.band_of(rank, world, H, p.yres, bounds)
        v_hi = min(b_hi, H)
        cols = workload.generate(np, b_lo * W * M, v_hi * W * M, W, H, M, f_hi=f_hi, focus_dist=150.0,
                                 tan_half_fov=common.tan_half_fov(p), n_extra=len(kinds) - 1)
        visits, kv = capi.make_visits(cols, visits_per_pixel=M, pixels_per_row=W, pixel_y0=b_lo)
        eng = OracleEngine(lib, p, table, visits, cols, kinds)
        band = distributed.frame_step_bands(eng, dist, H, p.yres, bounds)
        assert band == (b_lo, b_hi)
        # rank 0 assembles the tiled result for the comparison
        per = p.xres * eng.stride
        parts = [torch.zeros_like(eng.accum) for _ in range(world)] if rank == 0 else None
        mask = torch.zeros_like(eng.accum); mask[b_lo * per:b_hi * per] = 1
        dist.gather(eng.accum * mask, parts, dst=0)
        imgs = []
        for a in range(len(kinds)):
            t = torch.from_numpy(eng.resolved[a]).contiguous()
            g = [torch.zeros_like(t) for _ in range(world)] if rank == 0 else None
            dist.gather(t, g, dst=0)
            if rank == 0:
                imgs.append(sum(g).numpy())
        if rank == 0:
            q.put((sum(parts).numpy().copy(), imgs))
    else:
        n_local = workload.frame_visit_count(W, H, M, world, rank)
        assert list(distributed.partition_rows(H, world, rank)) == list(range(rank, H, world))
        cols = workload.generate(np, 0, n_local, W, H, M, f_hi=f_hi, focus_dist=150.0,
                                 tan_half_fov=common.tan_half_fov(p), row_stride=world, row_offset=rank,
                                 n_extra=len(kinds) - 1)
        visits, kv = capi.make_visits(cols, visits_per_pixel=M, pixels_per_row=W, pixel_y0=rank, pixel_row_stride=world)
        eng = OracleEngine(lib, p, table, visits, cols, kinds)
        distributed.frame_step(eng, dist)
        if rank == 0:
            q.put((eng.accum.numpy().copy(), [r.copy() for r in eng.resolved]))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _two_rank_vs_single(orc, kinds, bands=False, world=2, bounds=None, f_hi=0.03):
    W, H, M, samples = 48, 32, 9, 24
    # single rank reference
    from pota_amd import distributed
    p, model, table, keep = common.po_setup(W, H, samples_override=samples)
    visits, cols = common.make_stream(p, W, H, M, f_hi, n_extra=len(kinds) - 1)
    eng = OracleEngine(orc, p, table, visits, cols, kinds)
    distributed.frame_step(eng, None)
    ref_acc, ref_img = eng.accum.numpy().copy(), [r.copy() for r in eng.resolved]
    assert ref_acc.reshape(eng.np, -1)[:, -1].sum() > 0

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, W, H, M, f_hi, samples, kinds, q, bands, bounds)) for r in range(world)]
    for pr in procs:
        pr.start()
    acc, img = q.get(timeout=300)
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    ref_rec, rec = ref_acc.reshape(eng.np, -1), acc.reshape(eng.np, -1)
    for a, kind in enumerate(kinds):
        ra, ga = ref_rec[:, 4 * a:4 * a + 4].reshape(-1), rec[:, 4 * a:4 * a + 4].reshape(-1)
        if kind:
            # closest-filtered: the same visit wins every pixel, values are copies
            assert np.array_equal(ga, ra)
            assert np.array_equal(img[a], ref_img[a])
            assert np.count_nonzero(ra) > 0
        else:
            # identical up to fp32 summation order of the cross-rank splats
            m = ra != 0
            assert np.array_equal(ga != 0, m)
            assert float(np.max(np.abs(ga[m] - ra[m]) / np.abs(ra[m]))) < 1e-5
            mi = ref_img[a] != 0
            assert float(np.max(np.abs(img[a][mi] - ref_img[a][mi]) / np.abs(ref_img[a][mi]))) < 1e-5


def test_two_rank_gloo_equals_single_rank(orc):
    _two_rank_vs_single(orc, [0])


def test_two_rank_gloo_closest_aovs(orc):
    """closest-filtered AOVs: min-reduce of the winner keys + owner gather + sum"""
    _two_rank_vs_single(orc, [0, 1, 0, 1])


def test_tiled_output_two_ranks(orc):
    """row bands + exchange of the rows touched outside the own band (frame_step_bands)"""
    _two_rank_vs_single(orc, [0], bands=True)


def test_tiled_output_three_ranks_closest_aovs(orc):
    """three bands (a middle band has two neighbours; 32 rows do not divide evenly), closest AOVs ride along"""
    _two_rank_vs_single(orc, [0, 1, 0], bands=True, world=3)


def test_tiled_output_unequal_bands(orc):
    """bands cut where distributed.rebalance() would put them: unequal heights, same frame"""
    _two_rank_vs_single(orc, [0, 1], bands=True, world=3, bounds=[0, 7, 21, 32])


def test_tiled_output_four_ranks_thin_bands_few_highlights(orc):
    """Two-row bands in the middle and a handful of highlights: draws cross several bands, some ranks touch only some
    of the others and some pairs exchange nothing at all -- every rank still posts exactly the sends and receives its
    peers expect (the forms travel in the all-gather), with a real backend underneath."""
    _two_rank_vs_single(orc, [0, 1], bands=True, world=4, bounds=[0, 2, 4, 20, 32], f_hi=0.004)


def test_tiled_output_without_highlights_exchanges_nothing(orc):
    """no redistributed visit anywhere: every rank's exchange list is empty and the step must not wait for anybody"""
    _two_rank_vs_single(orc, [0], bands=True, world=3, f_hi=0.0)


class SparseOracleEngine(OracleEngine):
    """... with the pixel-list form of the exchange (HipEngine.compact_rows / sparse_buffers / merge_sparse): an entry
    per pixel of the rows that holds anything, its index within those rows, its record and (closest AOVs) its key."""

    def entry_floats(self):
        return self.stride

    def compact_rows(self, lo, hi):
        p0, p1 = lo * self.p.xres, hi * self.p.xres
        cap = max(16, (p1 - p0) // 4)
        rec = self.rec[p0:p1]
        hold = rec.abs().sum(dim=1) != 0
        if self.zkey is not None:
            hold |= self.zkey[p0:p1] != -1
        sel = torch.nonzero(hold).flatten()
        n = int(sel.numel())
        idx = torch.zeros(cap, dtype=torch.int32)
        vals = torch.zeros(cap * self.stride, dtype=torch.float32)
        keys = torch.zeros(cap, dtype=torch.int64) if self.zkey is not None else None
        if n <= cap:
            idx[:n] = sel.to(torch.int32)
            vals[:n * self.stride] = rec[sel].flatten()
            if keys is not None:
                keys[:n] = self.zkey[p0:p1][sel]
        return n, cap, idx, vals, keys

    def sparse_buffers(self, n):
        return (torch.empty(n, dtype=torch.int32), torch.empty(n * self.stride, dtype=torch.float32),
                torch.empty(n, dtype=torch.int64) if self.zkey is not None else None)

    def merge_sparse(self, lo, hi, n, idx, vals, keys):
        pix = lo * self.p.xres + idx[:n].to(torch.int64)
        assert int(pix.max()) < hi * self.p.xres and pix.unique().numel() == n
        src = vals[:n * self.stride].view(n, self.stride)
        closer = None
        if keys is not None:
            mine = self.zkey[pix].numpy().view(np.uint64)
            theirs = keys[:n].numpy().view(np.uint64)
            closer = torch.from_numpy(theirs < mine)
        for a in range(self.n_aovs):
            cols_a = slice(4 * a, 4 * a + 4)
            if self.kinds[a]:
                upd = self.rec[pix, cols_a]
                upd[closer] = src[:, cols_a][closer]
                self.rec[pix, cols_a] = upd
            else:
                self.rec[pix, cols_a] += src[:, cols_a]
        self.rec[pix, 4 * self.n_aovs] += src[:, 4 * self.n_aovs]
        if keys is not None:
            self.zkey[pix] = torch.from_numpy(np.minimum(mine, theirs).view(np.int64))


class _ThreadDist:
    """torch.distributed stand-in between engines that live in one process, one thread per rank: the calls
    frame_step_bands makes, with the semantics of a real backend -- the all-gather is collective, sends and receives
    pair up per (source, destination) in posting order and involve nobody else."""

    class _Req:
        def wait(self):
            return None

    class Shared:
        def __init__(self, world):
            import queue
            import threading
            self.world = world
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world
            self.mail = {(a, b): queue.Queue() for a in range(world) for b in range(world)}

    isend, irecv = "isend", "irecv"

    @staticmethod
    def P2POp(op, tensor, peer):
        return (op, tensor, peer)

    def __init__(self, shared, rank):
        self.sh, self.rank = shared, rank

    def get_world_size(self):
        return self.sh.world

    def get_rank(self):
        return self.rank

    def all_gather(self, out, t):
        self.sh.slots[self.rank] = t.clone()
        self.sh.barrier.wait(timeout=60)
        for k in range(self.sh.world):
            out[k].copy_(self.sh.slots[k])
        self.sh.barrier.wait(timeout=60)

    def batch_isend_irecv(self, ops):
        for op, t, peer in ops:
            if op == "isend":
                self.sh.mail[(self.rank, peer)].put(t.clone())
        for op, t, peer in ops:
            if op == "irecv":
                t.copy_(self.sh.mail[(peer, self.rank)].get(timeout=60))
        return [self._Req() for _ in ops]


def test_tiled_step_seeded_soak(orc):
    """Seeded soak of frame_step_bands: 2-5 ranks, even or random band boundaries down to two-row bands (draws cross
    several bands; some ranks exchange with some of the others only, or with nobody), frame size, highlight
    fraction, AOV kinds.  Every band against the same rows of a single-rank pass over the whole frame."""
    import threading
    from pota_amd import capi, distributed, workload
    n_cases = int(os.environ.get("LENTIL_SOAK_CASES", "10"))
    rng = np.random.default_rng(int(os.environ.get("LENTIL_SOAK_SEED", "0x7D1E"), 0))
    M = 9
    seen = {"lists": 0, "rows": 0, "nothing": 0}
    for case in range(n_cases):
        world = int(rng.integers(2, 6))
        W, H = int(rng.integers(24, 64)), int(rng.integers(4 * world, 44))
        cuts = sorted(rng.choice(np.arange(2, H - 1, 2), size=world - 1, replace=False).tolist())
        bounds = [0] + [int(c) for c in cuts] + [H] if rng.integers(0, 2) else None
        f_hi = float(rng.choice([0.0, 0.0015, 0.01, 0.03]))
        kinds = [[0], [0, 0], [0, 1, 0]][int(rng.integers(0, 3))]
        Engine = SparseOracleEngine if rng.integers(0, 3) else OracleEngine          # pixel lists where they fit / rows only
        tag = "case %d: world %d %dx%d bounds %r f_hi %g kinds %r %s" % (case, world, W, H, bounds, f_hi, kinds, Engine.__name__)
        p, model, table, keep = common.po_setup(W, H, samples_override=16)
        visits, cols = common.make_stream(p, W, H, M, f_hi, n_extra=len(kinds) - 1)
        whole = OracleEngine(orc, p, table, visits, cols, kinds)
        distributed.frame_step(whole, None)
        engines, keepalive, bands = [], [], []
        for rank in range(world):
            b_lo, b_hi = distributed.band_of(rank, world, H, p.yres, bounds)
            c = workload.generate(np, b_lo * W * M, min(b_hi, H) * W * M, W, H, M, f_hi=f_hi, focus_dist=150.0,
                                  tan_half_fov=common.tan_half_fov(p), n_extra=len(kinds) - 1)
            v, kv = capi.make_visits(c, visits_per_pixel=M, pixels_per_row=W, pixel_y0=b_lo)
            keepalive.append((c, v, kv))
            engines.append(Engine(orc, p, table, v, c, kinds))
            bands.append((b_lo, b_hi))
        shared, errors = _ThreadDist.Shared(world), []

        def run(rank):
            try:
                assert distributed.frame_step_bands(engines[rank], _ThreadDist(shared, rank), H, p.yres, bounds) == bands[rank]
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
            for f in distributed.LAST_FORMS[rank]:
                seen["lists" if f > 0 else ("rows" if f < 0 else "nothing")] += 1
        for rank in range(world):
            lo, hi = bands[rank][0] * p.xres, bands[rank][1] * p.xres
            got, ref = engines[rank].rec[lo:hi].numpy(), whole.rec[lo:hi].numpy()
            for a, kind in enumerate(kinds):
                g, r = got[:, 4 * a:4 * a + 4], ref[:, 4 * a:4 * a + 4]
                gi, ri = engines[rank].resolved[a][lo:hi], whole.resolved[a][lo:hi]
                if kind:
                    assert np.array_equal(g, r) and np.array_equal(gi, ri), "%s rank %d aov %d" % (tag, rank, a)
                else:
                    m = r != 0
                    assert np.array_equal(g != 0, m), "%s rank %d aov %d" % (tag, rank, a)
                    if m.any():
                        assert float(np.max(np.abs(g[m] - r[m]) / np.abs(r[m]))) < 1e-5, "%s rank %d aov %d" % (tag, rank, a)
                    mi = ri != 0
                    if mi.any():
                        assert float(np.max(np.abs(gi[mi] - ri[mi]) / np.abs(ri[mi]))) < 2e-5, "%s rank %d aov %d" % (tag, rank, a)
            gw, rw = got[:, -1], ref[:, -1]
            assert np.array_equal(gw != 0, rw != 0), tag
            assert float(np.max(np.abs(gw - rw) / np.maximum(np.abs(rw), 1e-30))) < 1e-5, tag
    if n_cases >= 10:
        assert all(seen.values()), seen            # every form of the exchange came up


def test_rebalance_keeps_tiny_frames_cuttable():
    """Frames with fewer rows per rank than min_rows: the boundaries stay strictly ascending inside the frame (the
    clamp used to push them past the end and back below zero), whatever the timings say."""
    from pota_amd import distributed
    rng = np.random.default_rng(5)
    for world, rows in [(4, 20), (8, 9), (2, 3), (5, 37), (8, 64), (3, 3)]:
        even = distributed.even_bounds(world, rows)
        for _ in range(20):
            sec = rng.uniform(1e-4, 1.0, world).tolist()
            new = distributed.rebalance(even, sec)
            assert new[0] == 0 and new[-1] == rows and len(new) == world + 1
            assert all(new[k] < new[k + 1] for k in range(world)), (world, rows, sec, new)
            again = distributed.rebalance(new, sec, damping=0.6)
            assert all(again[k] < again[k + 1] for k in range(world)) and again[0] == 0 and again[-1] == rows
    assert distributed.rebalance([0, 5, 10, 15, 20], [1.0, 1.0, 1.0, 1.0]) == [0, 5, 10, 15, 20]


def test_rebalance_equalises_the_modelled_cost():
    from pota_amd import distributed
    H, G = 6112, 8
    even = distributed.even_bounds(G, H)
    assert even[0] == 0 and even[-1] == H and len(even) == G + 1
    assert distributed.band_of(3, G, H, H + 1) == distributed.band_of(3, G, H, H + 1, even)
    assert distributed.band_of(G - 1, G, H, H + 1, even)[1] == H + 1       # the last band owns the extra frame row
    times = [2.99, 2.45, 2.51, 2.48, 2.43, 2.50, 2.40, 3.50]                 # measured: outer bands cost more
    new = distributed.rebalance(even, times)
    assert new[0] == 0 and new[-1] == H and all(b > a for a, b in zip(new, new[1:]))
    assert new[1] - new[0] < 764 and new[-1] - new[-2] < 764               # outer bands shrink
    dens = [times[i] / (even[i + 1] - even[i]) for i in range(G)]

    def cost(lo, hi):
        return sum(dens[i] * max(0, min(hi, even[i + 1]) - max(lo, even[i])) for i in range(G))

    c = [cost(new[i], new[i + 1]) for i in range(G)]
    assert max(c) / min(c) < 1.01
    assert distributed.rebalance(even, [1.0] * G) == even                   # balanced already: unchanged
    half = distributed.rebalance(even, times, damping=0.5)
    assert all(abs(h - e) <= abs(n - e) for h, n, e in zip(half, new, even))
    # degenerate input keeps every band at least min_rows high
    tiny = distributed.rebalance([0, 10, 20, 30], [100.0, 1e-9, 1e-9], min_rows=4)
    assert tiny[0] == 0 and tiny[-1] == 30 and all(b - a >= 4 for a, b in zip(tiny, tiny[1:]))


def test_modelled_bounds_level_the_measured_band_times():
    """The cut the calibration starts from (distributed.modelled_bounds): with the cost model's own numbers -- a row near the
    frame's top or bottom edge costs up to 2.4 rows -- the eight bands of the 4K frame carry equal modelled cost, the outer
    ones are the shortest, and the model reproduces the measured ratio of an outer to a middle even band (0.89 / 0.68 ms,
    profiles/r04_emulated_bands.txt)."""
    from pota_amd import distributed
    H, G = 2160, 8
    b = distributed.modelled_bounds(G, H)
    assert b[0] == 0 and b[-1] == H and len(b) == G + 1 and all(y > x for x, y in zip(b, b[1:]))
    er, w = 0.055 * H, 1.4
    cost = [1.0 + w * max(0.0, 1.0 - min(y + 0.5, H - y - 0.5) / er) for y in range(H)]
    per_band = [sum(cost[b[k]:b[k + 1]]) for k in range(G)]
    assert max(per_band) / min(per_band) < 1.02
    heights = [b[k + 1] - b[k] for k in range(G)]
    assert heights[0] == min(heights) and heights[-1] == min(heights[-1], heights[0] + 1) and heights[3] > heights[0]
    even = distributed.even_bounds(G, H)
    outer, middle = sum(cost[even[0]:even[1]]), sum(cost[even[3]:even[4]])
    assert abs(outer / middle - 0.89 / 0.68) < 0.03
    # symmetric, deterministic, and sane where there is nothing to model
    assert [H - x for x in reversed(b)] == b or max(abs((H - x) - y) for x, y in zip(reversed(b), b)) <= 1
    assert distributed.modelled_bounds(1, H) == [0, H]
    assert distributed.modelled_bounds(4, 6) == distributed.even_bounds(4, 6)
    for world, rows in [(2, 9), (8, 64), (3, 100), (8, 4320)]:
        m = distributed.modelled_bounds(world, rows)
        assert m[0] == 0 and m[-1] == rows and all(y > x for x, y in zip(m, m[1:]))


def test_row_partition_covers_frame():
    from pota_amd import distributed, workload
    H = 37
    for G in (1, 2, 3, 8):
        rows = sorted(r for k in range(G) for r in distributed.partition_rows(H, G, k))
        assert rows == list(range(H))
        assert sum(workload.frame_visit_count(10, H, 9, G, k) for k in range(G)) == 10 * H * 9
