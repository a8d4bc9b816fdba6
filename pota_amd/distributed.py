"""Multi-GPU frame redistribution: one process per GPU, visits partitioned by source-pixel row
(row r -> rank r mod G), every rank splats into its own full-frame accumulators, one sum
all-reduce (RCCL over xGMI; `nccl` backend of torch.distributed) merges the cross-tile splats, then
every rank resolves locally.  SURVEY.md section 8(e); the reference has no counterpart (single
process, threads sharing one set of buffers, src/lentil.h:823-851).

The step logic is engine-agnostic so that the N>1 path is covered by world_size-2 gloo tests on
CPU (tests/test_multi_gpu.py); the product engine is HipEngine (liblentil_hip.so).
"""
import ctypes as C
import os

FORCE_COLLECTIVE = os.environ.get("LENTIL_FORCE_DIST") == "1"


class _CudaArrayView:
    """Zero-copy view of a device allocation for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr, n_floats):
        self.__cuda_array_interface__ = {
            "shape": (int(n_floats),), "typestr": "<f4", "data": (int(ptr), False), "version": 2, "strides": None,
        }


class HipEngine:
    """Adapter: capi.Context -> the interface frame_step() drives."""

    def __init__(self, ctx):
        import torch
        self.ctx = ctx
        ptr, n = ctx.accum_buffer()
        self._view = _CudaArrayView(ptr, n)
        self.accum = torch.as_tensor(self._view, device=torch.device("cuda", torch.cuda.current_device()))
        assert self.accum.data_ptr() == ptr and self.accum.numel() == n

    def clear(self):
        self.ctx.clear_frame()

    def redistribute(self):
        self.ctx.redistribute()

    def finish_local(self):
        self.ctx.sync()              # accumulators complete before the collective reads them

    def before_resolve(self):
        import torch
        torch.cuda.synchronize()     # collective complete before the resolve kernel (library stream)

    def resolve(self):
        self.ctx.resolve()


def partition_rows(height, world_size, rank):
    """Rows owned by `rank`: r, r+G, r+2G, ... (interleaved for highlight load balance)."""
    return range(rank, height, world_size)


def frame_step(engine, dist=None):
    """One redistribution pass over the rank's visits incl. the cross-rank merge and the resolve."""
    engine.clear()
    engine.redistribute()
    if dist is not None and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVE):
        engine.finish_local()
        dist.all_reduce(engine.accum, op=dist.ReduceOp.SUM)
        engine.before_resolve()
    engine.resolve()
