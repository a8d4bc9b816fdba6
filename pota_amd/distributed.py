"""Multi-GPU frame redistribution: one process per GPU, visits partitioned by source-pixel row
(row r -> rank r mod G), every rank splats into its own full-frame accumulators, one sum
all-reduce (RCCL over xGMI; `nccl` backend of torch.distributed) merges the cross-tile splats, then
every rank resolves locally.  SURVEY.md section 8(e); the reference has no counterpart (single
process, threads sharing one set of buffers, src/lentil.h:823-851).

Frames with closest-filtered AOVs (one z-buffer for all of them, src/lentil.h:832-837) add one exchange
before the sum: the per-pixel winner keys (|Z| bits, then the later frame-wide visit id) are min-reduced,
every rank then writes the values of the winners it owns into otherwise zero accumulator slots, and the
sum all-reduce carries them to everybody.

Tiled output (`frame_step_bands`, the default of bench.py for N > 1): rank r owns a band of consecutive
rows -- its visits and its part of the output.  A draw lands within a bokeh radius of its visit, so
what a rank adds outside its band is confined to a few rows next to it.  After the pass the ranks tell
each other which rows they touched (one tiny all-gather), send exactly those foreign rows to their
owners (point-to-point over xGMI, neighbours only in practice) and each owner merges what it receives
(`lentil_hip_merge_rows`: sums, and the smaller winner key for closest-filtered AOVs) and resolves its
band.  The exchanged volume is 2 x reach x row size per rank instead of the whole frame.

The step logic is engine-agnostic so that the N>1 path is covered by world_size-2 gloo tests on
CPU (tests/test_multi_gpu.py); the product engine is HipEngine (liblentil_hip.so).
"""
import ctypes as C
import os

FORCE_COLLECTIVE = os.environ.get("LENTIL_FORCE_DIST") == "1"

# LENTIL_BAND_TIMING=1: host wall time per phase of frame_step_bands, accumulated here (development aid; the
# phases end with the synchronisations the step has anyway)
PHASE_SECONDS = {}
SPARSE_EXCHANGE = os.environ.get("LENTIL_EXCHANGE", "") != "rows"      # LENTIL_EXCHANGE=rows: always whole rows
LAST_FORMS = {}        # rank -> what frame_step_bands last sent to every rank (n > 0 entries, -1 whole rows, 0 nothing); for tests
_TIMING = os.environ.get("LENTIL_BAND_TIMING") == "1"


class _Phase:
    def __init__(self):
        import time
        self._now = time.perf_counter
        self.t = self._now()

    def mark(self, name):
        if _TIMING:
            t = self._now()
            PHASE_SECONDS[name] = PHASE_SECONDS.get(name, 0.0) + (t - self.t)
            self.t = t


class _CudaArrayView:
    """Zero-copy view of a device allocation for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr, n, typestr="<f4"):
        self.__cuda_array_interface__ = {
            "shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2, "strides": None,
        }


class HipEngine:
    """Adapter: capi.Context -> the interface frame_step() drives."""

    def __init__(self, ctx, rows=None):
        import torch
        self.ctx = ctx
        self.rows = rows             # yres of the frame (needed for the row views of the tiled mode)
        dev = torch.device("cuda", torch.cuda.current_device())
        self.device = dev
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

    # ---- tiled output: row views of the accumulator block / key buffer, merge, band resolve
    def touched_rows(self):
        return self.ctx.touched_rows()

    def acc_rows(self, lo, hi):
        per_row = self.accum.numel() // self.rows
        return self.accum[lo * per_row:hi * per_row]

    def key_rows(self, lo, hi):
        per_row = self.zkey.numel() // self.rows
        return self.zkey[lo * per_row:hi * per_row]

    def merge_rows(self, lo, acc, keys):
        per_row = self.accum.numel() // self.rows
        self.ctx.merge_rows(lo, acc.numel() // per_row, acc.data_ptr(), keys.data_ptr() if keys is not None else None)

    # the same exchange without the padding of the pixel records (5 instead of 8 floats per pixel, beauty only)
    def packed_row_floats(self):
        return int(self.ctx.params.xres) * (4 * self.ctx.n_aovs + 1)

    def pack_rows(self, lo, hi):
        import torch
        out = torch.empty((hi - lo) * self.packed_row_floats(), dtype=torch.float32, device=self.device)
        self.ctx.pack_rows(lo, hi - lo, out.data_ptr())
        return out

    def merge_packed_rows(self, lo, n_rows, packed, keys):
        self.ctx.merge_packed_rows(lo, n_rows, packed.data_ptr(), keys.data_ptr() if keys is not None else None)

    # ... and for rows that are mostly empty: one entry per pixel that holds anything
    def entry_floats(self):
        return 4 * self.ctx.n_aovs + 1

    def compact_rows(self, lo, hi):
        """-> (count, idx, vals, keys); count > capacity means "too dense, send the rows whole" (nothing usable written)."""
        import torch
        n_pix = (hi - lo) * int(self.ctx.params.xres)
        cap = max(1024, n_pix // 4)
        idx = torch.empty(cap, dtype=torch.int32, device=self.device)
        vals = torch.empty(cap * self.entry_floats(), dtype=torch.float32, device=self.device)
        keys = torch.empty(cap, dtype=torch.int64, device=self.device) if self.zkey is not None else None
        n = self.ctx.compact_rows(lo, hi - lo, idx.data_ptr(), vals.data_ptr(), keys.data_ptr() if keys is not None else None, cap)
        return n, cap, idx, vals, keys

    def sparse_buffers(self, n):
        import torch
        idx = torch.empty(n, dtype=torch.int32, device=self.device)
        vals = torch.empty(n * self.entry_floats(), dtype=torch.float32, device=self.device)
        keys = torch.empty(n, dtype=torch.int64, device=self.device) if self.zkey is not None else None
        return idx, vals, keys

    def merge_sparse(self, lo, hi, n, idx, vals, keys):
        self.ctx.merge_sparse(lo, hi - lo, n, idx.data_ptr(), vals.data_ptr(), keys.data_ptr() if keys is not None else None)

    def resolve_rows(self, lo, hi):
        self.ctx.resolve_rows(lo, hi - lo)

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

    def fold_direct(self):
        self.ctx.accum_buffer()      # what the scan keeps apart for the resolve goes into the block the collective sums

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
        if hasattr(engine, "fold_direct"):
            engine.fold_direct()
        engine.finish_local()
        if zkey is not None:
            exchange_closest(zkey, dist)
            engine.after_key_exchange()
            engine.closest_gather()
            engine.finish_local()
        dist.all_reduce(engine.accum, op=dist.ReduceOp.SUM)
        engine.before_resolve()
    engine.resolve()


def native_comm_init(ctx, dist):
    """Give `ctx` (capi.Context) the library's own RCCL communicator over the ranks of `dist`'s default group: rank 0
    draws the id, the group (any backend) carries its 128 bytes.  Collective."""
    from . import capi
    world, rank = dist.get_world_size(), dist.get_rank()
    box = [capi.Context.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    ctx.comm_init(box[0], rank, world)


def frame_step_native(ctx):
    """frame_step with the exchange inside liblentil_hip.so (lentil_hip_allreduce): clear, pass, min-reduce of the
    winner keys + gather (frames with closest AOVs), sum all-reduce, resolve -- all enqueued on the library's stream,
    no torch.distributed call on the data path."""
    ctx.set_closest_exchange(True)            # (means nothing for frames without closest-filtered AOVs)
    ctx.clear_frame()
    ctx.redistribute()
    ctx.allreduce()
    ctx.resolve()


def frame_step_bands_native(ctx, visit_rows, bounds=None):
    """frame_step_bands with the exchange inside liblentil_hip.so (lentil_hip_exchange_bands).  Returns the band."""
    ctx.set_closest_exchange(False)           # local winners are gathered by the pass; keys travel with the rows
    ctx.clear_frame()
    ctx.redistribute()
    return ctx.exchange_bands(visit_rows, bounds, sparse=SPARSE_EXCHANGE)


def even_bounds(world, visit_rows):
    """Band boundaries (world + 1 visit rows) of the even split."""
    return [visit_rows * r // world for r in range(world + 1)]


def modelled_bounds(world, visit_rows, edge_rows=None, edge_weight=1.4):
    """Band boundaries from a cost model instead of the even split -- where the calibration passes START, so that the first
    cut is already near the level one (round 5).  The model: a row costs 1, plus
    edge_weight * (1 - d / edge_rows) within edge_rows of the frame's top or bottom edge (d = rows to that edge).  Rows
    near an edge hold the items whose circle of confusion reaches over it: they lose attempts to the outside of the
    frame and trace at the largest field angles (slower Newton solves).  Measured on one MI355X running single bands of
    the 4K frame (profiles/r04_emulated_bands.txt): 0.89 ms for an outer band of 270 rows against 0.68 ms for a middle
    one -- the outer band carries 84 rows' worth of extra cost, which edge_rows = 5.5 % of the frame's height (120 rows
    at 2160) and edge_weight = 1.4 reproduce.  Pure function: every rank computes the same cut."""
    n = int(visit_rows)
    if world <= 1 or n < 2 * world:
        return even_bounds(world, n)
    er = float(edge_rows) if edge_rows else max(1.0, 0.055 * n)
    # cumulative cost in closed form per row is not worth it: a few thousand rows, summed once
    cost = [1.0 + edge_weight * max(0.0, 1.0 - min(y + 0.5, n - y - 0.5) / er) for y in range(n)]
    total = sum(cost)
    bounds, acc, y = [0], 0.0, 0
    for k in range(1, world):
        target = total * k / world
        while y < n and acc + cost[y] <= target:
            acc += cost[y]
            y += 1
        bounds.append(y)
    bounds.append(n)
    for k in range(1, world + 1):        # (every band at least one row)
        bounds[k] = max(bounds[k], bounds[k - 1] + 1)
    for k in range(world - 1, 0, -1):
        bounds[k] = min(bounds[k], bounds[k + 1] - 1)
    return bounds


def band_of(rank, world, visit_rows, frame_rows, bounds=None):
    """Rows [lo, hi) of the frame owned by `rank`: consecutive visit rows, split evenly or at `bounds` (world + 1
    ascending rows, bounds[0] = 0, bounds[world] = visit_rows); the last band also owns the rows beyond the visits
    (the reference allocates yres = H + 1, src/lentil.h:1069-1080)."""
    if bounds is None:
        lo = visit_rows * rank // world
        hi = visit_rows * (rank + 1) // world
    else:
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    if rank == world - 1:
        hi = frame_rows
    return lo, hi


def rebalance(bounds, seconds, min_rows=8, damping=1.0):
    """New band boundaries from the time every rank's pass took with the current ones: the cost of a row is taken as
    constant within a band (seconds[r] / rows of band r), the cumulative cost is cut into equal parts.  The field
    angle, and with it the cost of a backward trace, grows towards the frame edges, so even bands leave the outer
    ranks with up to 40 % more work.  Pure function of its arguments: every rank computes the same result from the
    all-gathered times.  `damping` < 1 moves only part of the way (the model is crude where the cost varies inside a
    band)."""
    world = len(bounds) - 1
    rows = [max(1, int(bounds[r + 1]) - int(bounds[r])) for r in range(world)]
    cost = [max(float(seconds[r]), 1e-9) for r in range(world)]
    total = sum(cost)
    new = [int(bounds[0])]
    r, acc = 0, 0.0                      # acc: cost of the bands before band r
    for k in range(1, world):
        target = total * k / world
        while r < world - 1 and acc + cost[r] < target:
            acc += cost[r]
            r += 1
        frac = (target - acc) / cost[r]
        row = bounds[r] + frac * rows[r]
        row = bounds[k] + damping * (row - bounds[k])
        new.append(int(round(row)))
    new.append(int(bounds[world]))
    # a band never shrinks below half or grows beyond twice the even height, whatever the timings say (the physical
    # imbalance is a few tens of percent; a rank that was merely disturbed while it was timed must not wreck the cut)
    even = (int(bounds[world]) - int(bounds[0])) / world
    lo_h = min(max(min_rows, int(even * 0.5)), int(even))      # never more than the even height: the bands must fit
    hi_h = max(min_rows, int(even * 2.0), lo_h)
    for k in range(1, world):            # boundaries ascending, heights within [lo_h, hi_h] ...
        new[k] = min(max(new[k], new[k - 1] + lo_h), new[k - 1] + hi_h)
    for k in range(world - 1, 0, -1):    # ... also seen from the other end (the last boundary is fixed)
        new[k] = max(min(new[k], new[k + 1] - lo_h), new[k + 1] - hi_h)
    ok = all(new[k] < new[k + 1] for k in range(world)) and new[0] == bounds[0] and new[world] == bounds[world]
    return new if ok else [int(b) for b in bounds]      # frames too small to cut any other way keep their bounds


def frame_step_bands(engine, dist, visit_rows, frame_rows, bounds=None):
    """One pass with tiled output: redistribute the rank's band, exchange the rows touched outside the own band with
    their owners, merge what arrives, resolve the own band.  Returns the band.  `bounds`: see band_of (the engine's
    visits must be those of band_of(rank, ..., bounds))."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    zkey = getattr(engine, "zkey", None)
    if zkey is not None:
        engine.set_deferred_closest(False)        # local winners are gathered by the pass; keys travel with the rows
    ph = _Phase()
    engine.clear()
    engine.redistribute()
    ph.mark("enqueue pass")
    engine.finish_local()
    ph.mark("pass done")
    band = band_of(rank, world, visit_rows, frame_rows, bounds)
    lo, hi = engine.touched_rows()
    packed = hasattr(engine, "pack_rows")          # rows travel without the record padding where the engine can
    sparse = SPARSE_EXCHANGE and hasattr(engine, "compact_rows")       # ... and as lists of the pixels that hold anything when few do
    # what this rank added to every other band: decided before the all-gather, which then also carries the form it
    # will arrive in (entries per destination: n > 0 a list of n pixels, -1 whole rows, 0 nothing)
    outgoing, forms = {}, [0] * world
    for q in range(world):
        if q == rank:
            continue
        q_lo, q_hi = band_of(q, world, visit_rows, frame_rows, bounds)
        s_lo, s_hi = max(lo, q_lo), min(hi, q_hi)                  # rows of q's band this rank added to
        if s_hi <= s_lo:
            continue
        forms[q] = -1
        if sparse:
            n, cap, idx, vals, keys = engine.compact_rows(s_lo, s_hi)
            if n <= cap:
                forms[q] = n
                if n:
                    outgoing[q] = ("sparse", n, idx, vals, keys)
                continue
        outgoing[q] = ("rows", s_lo, s_hi)
    LAST_FORMS[rank] = list(forms)
    ph.mark("compact foreign rows")
    # one small all-gather: staged through pinned memory, gathered into one tensor, read back with one copy
    cuda = torch.device(engine.device).type == "cuda"
    stage = torch.tensor([lo, hi] + forms, dtype=torch.int64, pin_memory=cuda)
    mine = stage.to(engine.device, non_blocking=True)
    gathered = torch.empty((world, 2 + world), dtype=torch.int64, device=engine.device)
    dist.all_gather(list(gathered.unbind(0)), mine)
    info = gathered.tolist()
    ph.mark("touched rows all-gather")
    ops, incoming, keep = [], [], []
    for q in range(world):
        if q == rank:
            continue
        out = outgoing.get(q)
        if out and out[0] == "sparse":
            _, n, idx, vals, keys = out
            ops.append(dist.P2POp(dist.isend, idx[:n], q))
            ops.append(dist.P2POp(dist.isend, vals[:n * engine.entry_floats()], q))
            if keys is not None:
                ops.append(dist.P2POp(dist.isend, keys[:n], q))
        elif out:
            _, s_lo, s_hi = out
            buf = engine.pack_rows(s_lo, s_hi) if packed else engine.acc_rows(s_lo, s_hi)
            keep.append(buf)
            ops.append(dist.P2POp(dist.isend, buf, q))
            if zkey is not None:
                ops.append(dist.P2POp(dist.isend, engine.key_rows(s_lo, s_hi), q))
        r_lo, r_hi = max(info[q][0], band[0]), min(info[q][1], band[1])   # rows of this band q added to
        form = info[q][2 + rank]
        if r_hi > r_lo and form > 0:
            idx, vals, keys = engine.sparse_buffers(form)
            ops.append(dist.P2POp(dist.irecv, idx, q))
            ops.append(dist.P2POp(dist.irecv, vals, q))
            if keys is not None:
                ops.append(dist.P2POp(dist.irecv, keys, q))
            incoming.append(("sparse", r_lo, r_hi, form, idx, vals, keys))
        elif r_hi > r_lo and form < 0:
            if packed:
                acc = torch.empty((r_hi - r_lo) * engine.packed_row_floats(), dtype=torch.float32, device=engine.device)
            else:
                acc = torch.empty_like(engine.acc_rows(r_lo, r_hi))
            keys = torch.empty_like(engine.key_rows(r_lo, r_hi)) if zkey is not None else None
            ops.append(dist.P2POp(dist.irecv, acc, q))
            if keys is not None:
                ops.append(dist.P2POp(dist.irecv, keys, q))
            incoming.append(("rows", r_lo, r_hi, acc, keys))
    if packed and keep:
        engine.finish_local()                      # the pack kernels ran on the library's stream
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    engine.before_resolve()                                        # received rows complete before the merge kernels
    ph.mark("row exchange")
    for item in incoming:
        if item[0] == "sparse":
            _, r_lo, r_hi, n, idx, vals, keys = item
            engine.merge_sparse(r_lo, r_hi, n, idx, vals, keys)
        else:
            _, r_lo, r_hi, acc, keys = item
            if packed:
                engine.merge_packed_rows(r_lo, r_hi - r_lo, acc, keys)
            else:
                engine.merge_rows(r_lo, acc, keys)
    engine.resolve_rows(band[0], band[1])
    if incoming:
        engine.finish_local()      # the merge kernels read torch-owned buffers: done before those are released
    ph.mark("merge + resolve enqueue")
    return band
