"""Multi-GPU frame redistribution: one process per GPU, visits partitioned by source-pixel row
(row r -> rank r mod G), every rank splats into its own full-frame accumulators, one sum
all-reduce (RCCL over xGMI; `nccl` backend of torch.distributed) merges the cross-tile splats, then
every rank resolves locally.  SURVEY.md section 8(e); the reference has no counterpart (single
process, threads sharing one set of buffers, src/lentil.h:823-851).

Frames with closest-filtered AOVs (one z-buffer for all of them, src/lentil.h:832-837) add one exchange
before the sum: the per-pixel winner keys (|Z| bits, then the later frame-wide visit id) are min-reduced,
every rank then writes the values of the winners it owns into otherwise zero accumulator slots, and the
sum all-reduce carries them to everybody.

The step logic is engine-agnostic so that the N>1 path is covered by world_size-2 gloo tests on
CPU (tests/test_multi_gpu.py); the product engine is HipEngine (liblentil_hip.so).
"""
import ctypes as C
import os

FORCE_COLLECTIVE = os.environ.get("LENTIL_FORCE_DIST") == "1"


class _CudaArrayView:
    """Zero-copy view of a device allocation for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr, n, typestr="<f4"):
        self.__cuda_array_interface__ = {
            "shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2, "strides": None,
        }


class HipEngine:
    """Adapter: capi.Context -> the interface frame_step() drives."""

    def __init__(self, ctx):
        import torch
        self.ctx = ctx
        dev = torch.device("cuda", torch.cuda.current_device())
        ptr, n = ctx.accum_buffer()
        self._view = _CudaArrayView(ptr, n)
        self.accum = torch.as_tensor(self._view, device=dev)
        assert self.accum.data_ptr() == ptr and self.accum.numel() == n
        # closest-filtered AOVs: the winner keys as int64 (see exchange_closest)
        self.zkey = None
        kptr, kn = ctx.zkey_buffer()
        if kn:
            self._kview = _CudaArrayView(kptr, kn, "<i8")
            self.zkey = torch.as_tensor(self._kview, device=dev)
            assert self.zkey.data_ptr() == kptr and self.zkey.numel() == kn

    def set_deferred_closest(self, on):
        self.ctx.set_closest_exchange(on)

    def closest_gather(self):
        self.ctx.closest_gather()

    def after_key_exchange(self):
        import torch
        torch.cuda.synchronize()     # min-reduced keys complete before the gather kernel (library stream)

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


_SIGN = -(1 << 63)


def exchange_closest(zkey, dist):
    """Unsigned 64-bit minimum over ranks of the winner keys, with the signed MIN torch.distributed has:
    flipping the top bit maps unsigned order onto signed order (empty = all ones becomes INT64_MAX)."""
    zkey.bitwise_xor_(_SIGN)
    dist.all_reduce(zkey, op=dist.ReduceOp.MIN)
    zkey.bitwise_xor_(_SIGN)


def frame_step(engine, dist=None):
    """One redistribution pass over the rank's visits incl. the cross-rank merge and the resolve."""
    collective = dist is not None and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVE)
    zkey = getattr(engine, "zkey", None)
    if zkey is not None:
        engine.set_deferred_closest(collective)
    engine.clear()
    engine.redistribute()
    if collective:
        engine.finish_local()
        if zkey is not None:
            exchange_closest(zkey, dist)
            engine.after_key_exchange()
            engine.closest_gather()
            engine.finish_local()
        dist.all_reduce(engine.accum, op=dist.ReduceOp.SUM)
        engine.before_resolve()
    engine.resolve()
