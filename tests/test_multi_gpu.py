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
    """CPU stand-in for HipEngine with the same interface (clear/redistribute/accum/resolve, and the
    closest-AOV key exchange when the frame has closest-filtered AOVs)."""

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
        self.accum = torch.zeros(self.np * (4 * self.n_aovs + 1), dtype=torch.float32)
        self.zkey = torch.zeros(self.np, dtype=torch.int64) if any(self.kinds) else None
        self.deferred = False
        self.resolved = None

    def _aov(self, a):
        return self.accum[a * self.np * 4:(a + 1) * self.np * 4]

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
            self._aov(a)[:] = torch.from_numpy(self.frame.buffer(a).reshape(-1))
        self.accum[self.np * 4 * self.n_aovs:] = torch.from_numpy(self.frame.weight())
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
                dst = self._aov(a).reshape(-1, 4)
                dst[torch.from_numpy(mine)] = torch.from_numpy(self.cols["extra"][a - 1][v[mine]])

    def before_resolve(self):
        pass

    def resolve(self):
        w = self.accum[self.np * 4 * self.n_aovs:].numpy()
        self.resolved = []
        for a in range(self.n_aovs):
            acc = self._aov(a).reshape(-1, 4).numpy()
            out = acc.copy()
            if not self.kinds[a]:
                m = w != 0
                out[m] = acc[m] * (np.float32(1.0) / w[m])[:, None]
            self.resolved.append(out)


def _worker(rank, world, port, W, H, M, f_hi, samples, kinds, q):
    sys.path.insert(0, common.ROOT)
    sys.path.insert(0, os.path.join(common.ROOT, "tests"))
    import oracle_lib
    from pota_amd import capi, distributed, workload
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = oracle_lib.load()
    p, model, table, keep = common.po_setup(W, H, samples_override=samples)
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


def _two_rank_vs_single(orc, kinds):
    W, H, M, f_hi, samples = 48, 32, 9, 0.03, 24
    # single rank reference
    from pota_amd import distributed
    p, model, table, keep = common.po_setup(W, H, samples_override=samples)
    visits, cols = common.make_stream(p, W, H, M, f_hi, n_extra=len(kinds) - 1)
    eng = OracleEngine(orc, p, table, visits, cols, kinds)
    distributed.frame_step(eng, None)
    ref_acc, ref_img = eng.accum.numpy().copy(), [r.copy() for r in eng.resolved]
    assert ref_acc[-eng.np:].sum() > 0

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, W, H, M, f_hi, samples, kinds, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    acc, img = q.get(timeout=300)
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    n4 = eng.np * 4
    for a, kind in enumerate(kinds):
        ra, ga = ref_acc[a * n4:(a + 1) * n4], acc[a * n4:(a + 1) * n4]
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


def test_row_partition_covers_frame():
    from pota_amd import distributed, workload
    H = 37
    for G in (1, 2, 3, 8):
        rows = sorted(r for k in range(G) for r in distributed.partition_rows(H, G, k))
        assert rows == list(range(H))
        assert sum(workload.frame_visit_count(10, H, 9, G, k) for k in range(G)) == 10 * H * 9
