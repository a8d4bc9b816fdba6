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
    """CPU stand-in for HipEngine with the same interface (clear/redistribute/accum/resolve)."""

    def __init__(self, lib, p, table, visits):
        import ctypes as C
        import oracle_lib
        self.lib, self.p, self.visits = lib, p, visits
        self.lens = lib.orc_lens_create(C.byref(table))
        self.frame = None
        self.oracle_lib = oracle_lib
        self.np = p.xres * p.yres
        self.accum = torch.zeros(self.np * 5, dtype=torch.float32)
        self.resolved = None

    def clear(self):
        self.frame = self.oracle_lib.Frame(self.lib, self.p, n_aovs=1, shadow=False)
        self.accum.zero_()

    def redistribute(self):
        self.frame.run(self.lens, None, self.visits)
        self.accum[: self.np * 4] = torch.from_numpy(self.frame.buffer(0).reshape(-1))
        self.accum[self.np * 4:] = torch.from_numpy(self.frame.weight())

    def finish_local(self):
        pass

    def before_resolve(self):
        pass

    def resolve(self):
        acc = self.accum[: self.np * 4].reshape(-1, 4).numpy()
        w = self.accum[self.np * 4:].numpy()
        out = acc.copy()
        m = w != 0
        out[m] = acc[m] * (np.float32(1.0) / w[m])[:, None]
        self.resolved = out


def _worker(rank, world, port, W, H, M, f_hi, samples, q):
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
                             tan_half_fov=common.tan_half_fov(p), row_stride=world, row_offset=rank)
    visits, kv = capi.make_visits(cols, visits_per_pixel=M, pixels_per_row=W, pixel_y0=rank, pixel_row_stride=world)
    eng = OracleEngine(lib, p, table, visits)
    distributed.frame_step(eng, dist)
    if rank == 0:
        q.put((eng.accum.numpy().copy(), eng.resolved.copy()))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_rank_gloo_equals_single_rank(orc):
    W, H, M, f_hi, samples = 48, 32, 9, 0.03, 24
    # single rank reference
    from pota_amd import distributed
    p, model, table, keep = common.po_setup(W, H, samples_override=samples)
    visits, cols = common.make_stream(p, W, H, M, f_hi)
    eng = OracleEngine(orc, p, table, visits)
    distributed.frame_step(eng, None)
    ref_acc, ref_img = eng.accum.numpy().copy(), eng.resolved.copy()
    assert ref_acc[-eng.np:].sum() > 0

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, W, H, M, f_hi, samples, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    acc, img = q.get(timeout=300)
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    # identical up to fp32 summation order of the cross-rank splats
    m = ref_acc != 0
    assert np.array_equal(acc != 0, m)
    assert float(np.max(np.abs(acc[m] - ref_acc[m]) / np.abs(ref_acc[m]))) < 1e-5
    mi = ref_img != 0
    assert float(np.max(np.abs(img[mi] - ref_img[mi]) / np.abs(ref_img[mi]))) < 1e-5


def test_row_partition_covers_frame():
    from pota_amd import distributed, workload
    H = 37
    for G in (1, 2, 3, 8):
        rows = sorted(r for k in range(G) for r in distributed.partition_rows(H, G, k))
        assert rows == list(range(H))
        assert sum(workload.frame_visit_count(10, H, 9, G, k) for k in range(G)) == 10 * H * 9
