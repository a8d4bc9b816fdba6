#!/usr/bin/env python3
"""bench.py -- bidirectional redistribution throughput (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one redistribution pass over one frame's worth of synthetic AOV-sample visits that are
already resident in HBM: clear accumulators -> scan/compact/direct-accumulate -> draw/splat ->
(N>1: exchange of the cross-tile splats over xGMI) -> resolve.  `value` = visits consumed by all ranks
per second (Msamples/s; 1 sample = 1 visit record of 80+16K bytes).

Workload at N=1: the configuration the metric is quoted on -- double-gauss 50 mm polynomial optics,
3840x2160, 9 visits/pixel (AA 3), 1024 redistribution draws per redistributed visit, beauty only,
highlight fraction f_hi (default 2^-16, the scan-dominated regime of SURVEY.md section 8d; the
highlight-heavy regime 1.6e-3 is reported beside it in "regimes").  N>1 is the metric's wording -- the SAME 4K
frame at 1/2/4/8 GPUs (strong scaling, the default): rank r owns a band of consecutive rows -- its visits and its
tile of the output -- and sends what its draws touched outside the band to the owners over xGMI
(lentil_hip_exchange_bands; pota_amd/distributed.py::frame_step_bands is the torch.distributed form of the same
step).  At N>1 BASELINE config 5 (7680x4320, 2048 draws, tiled across the N GPUs) is timed the same way and
reported under "configs"; --scaling weak keeps the per-GPU work fixed instead (the same camera at N times the pixels).
LENTIL_PARTITION=interleaved selects rows r mod N with one sum all-reduce of the whole frame instead.

Launch: `python bench.py --gpus N` starts its own N ranks (a child `python -m torch.distributed.run ... bench.py`,
spawned before this process touches the GPU; its JSON line and exit code are passed on); under a launcher
(WORLD_SIZE set) it is one of the ranks.

The timed steps ALTERNATE between two resident visit streams generated from different seeds (different
highlights in different places): a pass sizes its buffers from what the previous pass found, and a renderer
never sees the same frame twice.  `passes` in the JSON line says how the timed passes ran (streamed / chunks
enqueued blind / redone because the estimate was too small).

Set-up before the W warm-up steps (untimed, like generating the visits): two passes that let the context size its
draw buffers (the first pass of a context waits for each chunk's scan and allocates); for N > 1 three calibration
passes that re-cut the row bands from the ranks' pass times.

The printed JSON line also carries
  roofline     -- the HBM-bound scan kernel: algorithmic bytes (visits x 80 B) / its mean duration,
                  measured with HIP events on the library's stream -- the start / stop events of the launch itself
                  (hipExtLaunchKernelGGL) in a streamed pass: events recorded *around* the launch read ~90 us more than
                  rocprofv3's kernel trace, the marker behind the scan waits for the command processor -- against 8 TB/s; `bytes_moved_per_visit` is
                  what the kernel actually requests (it does not read raydir_time unless a visit is at infinite
                  depth, and stores a record per pixel); `whole_step_frac` relates the frame's bytes to the whole step;
                  `frac` is the launch inside the timed pass, where the kernel shares its CUs with the solve waves that take
                  its output as it comes; `alone` is the same kernel with the chip to itself (the chunked form of the pass,
                  one chunk: LENTIL_STREAM=0 LENTIL_CHUNKS=1; N=1 only, --no-scan-alone skips it)
  parity_checked -- the timed frame (or every n-th row of it, sized for ~10 s of oracle work) through the HIP path and
                  through the oracle: accepted-draw lists bit for bit, radiance at 1e-5 (N=1 only)
  solve_fp64   -- the fp64-VALU-bound draw kernels: Newton lane-iterations x operations per iteration
                  (counted from the lens table) against the fp64 vector peak, per regime
  cpu_baseline -- the oracle (a port of the reference CPU path) timed on this box's host cores on a
                  bounded row sample of the same workload (rank 0, N=1 only): all threads and one thread
  configs      -- BASELINE.json's configs 2, 3 and 4 measured the same way (N=1 only; --no-configs skips them), each with a
                  parity_checked of its own (~5 s of oracle work on every n-th row of that config's frame).
"""
import argparse
import json
import math
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# A pass uses several HIP streams (scan, publishers, solve waves beside the scan); torch and RCCL bring theirs.
# With the runtime's default of 4 hardware queues per process two of them end up sharing one.
# Read when the HIP runtime initialises, so: before torch is imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# dmabuf IPC for RCCL between the ranks of one node (the host driver here supports nothing else); also read early
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_VECTOR_PEAK_TFLOPS = 78.6  # counts a fused multiply-add as two; the reference's arithmetic has none
SEEDS = (0x5EED, 0xBEEF)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--visits-per-pixel", type=int, default=9)
    ap.add_argument("--samples", type=int, default=1024)
    ap.add_argument("--f-hi", type=float, default=2.0 ** -16)
    ap.add_argument("--aovs", type=int, default=0, help="extra (non-beauty) AOVs")
    ap.add_argument("--lens", default="double_gauss_50mm")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-check", action="store_true", help="skip the oracle check of a sample of the workload")
    ap.add_argument("--no-scan-alone", action="store_true", help="skip the measurement of the scan kernel with the chip to itself")
    ap.add_argument("--no-pcie", action="store_true", help="skip the PCIe-inclusive measurement (upload from host memory + pass)")
    ap.add_argument("--no-second-regime", action="store_true")
    ap.add_argument("--no-sustained", action="store_true", help="skip roofline.sustained (350 untimed + 200 timed steps behind the timed ones)")
    ap.add_argument("--no-clock-trace", action="store_true", help="skip roofline.clock_in_pass (the sampler loop behind the timed steps)")
    ap.add_argument("--no-configs", action="store_true", help="skip BASELINE.json's configs 2-4")
    ap.add_argument("--same-frame", action="store_true", help="replay ONE visit stream (development aid; the default alternates two)")
    ap.add_argument("--cpu-row-step", type=int, default=0, help="CPU baseline samples every n-th row (0 = auto)")
    ap.add_argument("--bokeh-image", action="store_true", help="aperture draws from the reference's example bokeh image "
                                                               "(tests/golden/example_bokeh_kernel_u8.npy; BASELINE config 3)")
    ap.add_argument("--scaling", default="strong", choices=("strong", "weak"),
                    help="N > 1: strong = the same frame tiled over the GPUs (the metric's wording); weak = N times the pixels")
    ap.add_argument("--no-config5", action="store_true", help="N > 1: skip BASELINE config 5 (7680x4320, 2048 draws, tiled)")
    ap.add_argument("--bounds", default="", help="band boundaries (N+1 visit rows, comma separated) instead of the even split")
    ap.add_argument("--emulate", default="", help="development aid: 'N,r' runs rank r's band of the N-GPU frame in one "
                                                   "process (no exchange): per-band cost of the weak-scaling workload")
    return ap.parse_args()


def physical_cores():
    try:
        seen = set()
        phys = core = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":")[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        seen.add((phys, core))
                    phys = core = None
        return len(seen) or None
    except OSError:
        return None


def pcie_inclusive(b):
    """The same frame when its visit columns start in (page-locked) host memory: lentil_hip_upload_visits + the pass,
    best of two.  Never `value`: the metric is quoted with the inputs resident in HBM.  (A renderer hands the visits
    over while it renders -- lentil_hip_visits_append, tools/pcie_rate.py, profiles/r02_pcie_rate.json -- and then
    the frame end waits for the pass only.)"""
    import ctypes as C
    import numpy as np
    from pota_amd import capi
    torch = b.torch
    cols, v, kv = b.streams[0]
    names = ("rgba", "pos_z", "raydir_time", "volume_ignore", "transmission")
    n = int(v.n)
    nbytes = n * 16 * (5 + b.aovs)
    ptr = capi.host_alloc(nbytes)
    try:
        host = np.frombuffer((C.c_char * nbytes).from_address(ptr), np.float32).reshape(5 + b.aovs, n, 4)
        ht = torch.from_numpy(host)
        for c, name in enumerate(names):
            ht[c].copy_(cols[name])
        for k in range(b.aovs):
            ht[5 + k].copy_(cols["extra"][k])
        torch.cuda.synchronize()
        hcols = {name: host[c] for c, name in enumerate(names)}
        hcols["extra"] = [host[5 + k] for k in range(b.aovs)]
        hv, hkv = capi.make_visits(hcols, visits_per_pixel=b.M, pixels_per_row=b.W, pixel_y0=int(v.pixel_y0),
                                   pixel_row_stride=int(v.pixel_row_stride))
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            b.ctx.upload_visits(hv)
            b.ctx.sync()
            t1 = time.perf_counter()
            b.ctx.clear_frame(); b.ctx.redistribute(); b.ctx.resolve(); b.ctx.sync()
            t2 = time.perf_counter()
            if best is None or t2 - t0 < best[0]:
                best = (t2 - t0, t1 - t0, t2 - t1)
        del ht, host, hcols, hv, hkv
    finally:
        capi.host_free(ptr)
    b.ctx.bind_visits(v, kv)
    return {"value": round(n / best[0] / 1e6, 1), "unit": "Msamples/s", "upload_ms": round(best[1] * 1e3, 2),
            "upload_GBps": round(nbytes / best[1] / 1e9, 1), "pass_ms": round(best[2] * 1e3, 3),
            "what": "one lentil_hip_upload_visits of the whole frame from page-locked host memory + the pass; "
                    "handed over in blocks during the render instead, the frame end waits for the pass only "
                    "(profiles/r02_pcie_rate.json)"}


def cpu_baseline(args, p, table, M, tan_half_fov):
    """Oracle ("port" of the reference CPU path) on a bounded sample: every row_step-th image row of the same frame, on ALL of
    this box's logical CPUs (orc_redistribute_threads: the rows cut into one contiguous range per thread, one shared frame --
    a visit that stays in its pixel is added by the thread that owns the row, the accepted draws are kept as records and added
    in visit order at the end; no per-thread frames, so the thread count is not bounded by memory); then the same on one thread
    over a fraction of those rows."""
    import ctypes as C
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from pota_amd import capi, workload

    lib = oracle_lib.load()
    W, H = args.width, args.height
    logical = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))          # (a container may see fewer CPUs than the box has)
    except (AttributeError, OSError):
        usable = logical
    threads = max(1, min(usable, 512))

    def timed(n_threads, budget_s, repeats):
        # bound the CPU work: ~0.1 us per scanned visit, ~35 us per draw attempt per core
        est_full = W * H * M * (0.1e-6 + args.f_hi * args.samples * 35e-6 * 1.1)
        row_step = args.cpu_row_step or max(1, int(round(est_full / (budget_s * n_threads))))
        rows = list(range(0, H, row_step))
        n = len(rows) * W * M
        cols = workload.generate(np, 0, n, W, H, M, f_hi=args.f_hi, focus_dist=150.0, tan_half_fov=tan_half_fov,
                                 n_extra=args.aovs, row_stride=row_step, row_offset=0)
        visits, keep = capi.make_visits(cols, visits_per_pixel=M, pixels_per_row=W, pixel_row_stride=row_step)
        lens = lib.orc_lens_create(C.byref(table))
        best = None
        for _ in range(repeats):
            frame = oracle_lib.Frame(lib, p, n_aovs=1 + args.aovs, shadow=False)
            t0 = time.perf_counter()
            if n_threads == 1:
                frame.run(lens, None, visits)
            elif not frame.run_threads(lens, None, visits, n_threads, W * M):
                raise RuntimeError("orc_redistribute_threads does not take this workload")
            dt = time.perf_counter() - t0
            c = frame.counters()
            frame.close()
            if best is None or dt < best[0]:
                best = (dt, c)
        lib.orc_lens_destroy(lens)
        dt, c = best
        return n / dt / 1e6, "every %d-th row of the same %dx%d frame: %d visits, %d redistributed, %d draw attempts, %.2f s (best of %d)" % (
            row_step, W, H, n, c.redistributed_visits, c.attempted_draws, dt, repeats)

    v_all, s_all = timed(threads, 20.0, 3)
    v_one, s_one = timed(1, 6.0, 1)
    return {
        "value": round(v_all, 4), "unit": "Msamples/s", "cores": threads, "kind": "port", "sample": s_all,
        "threads": threads, "logical_cpus": logical, "usable_cpus": usable, "physical_cores": physical_cores(),
        "one_thread": {"value": round(v_one, 4), "unit": "Msamples/s", "sample": s_one},
    }


def parity_check(W, H, M, samples, aovs, f_hi, p, table, tan_half_fov, device_index, torch, bokeh_tables=None, budget_s=10.0,
                 full_size=None):
    """The same check as tests/ make, inside the bench run: a bounded sample of a timed workload (every row_step-th
    image row of the same frame, ~budget_s of oracle work) goes through the HIP path and through the oracle (fp32
    buffers as the reference keeps them, fp64 shadows beside them); accepted draws are compared as (visit, attempt,
    pixel) lists, bit for bit, radiance at 1e-5.  The timed passes themselves are checked at full size by
    tests/test_gpu_headline.py (a minute or more of oracle work each: not inside a bench run)."""
    import ctypes as C
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    import common
    from test_gpu_parity import check_frame, check_logs
    from pota_amd import _abi, capi, workload

    lib = oracle_lib.load()
    n_aovs = 1 + aovs
    # (one shared oracle frame whatever the thread count: orc_redistribute_threads)
    threads = max(1, min(os.cpu_count() or 1, 64))
    est_full = W * H * M * (0.1e-6 + f_hi * samples * 35e-6 * 1.1) * 1.5       # (shadow buffers, the draw log)
    row_step = max(1, int(round(est_full / (budget_s * threads))))
    rows = list(range(0, H, row_step))
    n = len(rows) * W * M
    cols = workload.generate(np, 0, n, W, H, M, f_hi=f_hi, focus_dist=150.0, tan_half_fov=tan_half_fov,
                             n_extra=aovs, row_stride=row_step, row_offset=0)
    visits, keep = capi.make_visits(cols, visits_per_pixel=M, pixels_per_row=W, pixel_row_stride=row_step)
    ob = None
    if bokeh_tables is not None:
        bt = _abi.BokehTable()
        bt.x, bt.y = bokeh_tables["x"], bokeh_tables["y"]
        for k in ("cdfRow", "rowIndices", "cdfColumn", "columnIndices"):
            setattr(bt, k, bokeh_tables[k].ctypes.data)
        ob = lib.orc_bokeh_from_tables(C.byref(bt))
    ref = common.ThreadedOracle(lib, p, table, visits, n_threads=threads, n_aovs=n_aovs, row_visits=W * M, bokeh=ob)
    ctx = capi.Context(device_index)
    try:
        ctx.set_params(p); ctx.set_lens(table); ctx.set_bokeh(bokeh_tables); ctx.alloc_frame(n_aovs); ctx.set_draw_log(1 << 24)
        ctx.upload_visits(visits)
        ctx.clear_frame(); ctx.redistribute(); ctx.resolve(); ctx.sync()
        c = ctx.counters()
        check_logs(ctx, ref)                       # AssertionError: caught by the caller, reported as ok = false
        st = {}
        worst = check_frame(ctx, ref, n_aovs=n_aovs, tol=1e-5, stats=st)
        n_log = int(ctx.draw_log().shape[0])
    finally:
        ctx.close()
        ref.close()
        if ob:
            lib.orc_bokeh_destroy(ob)
    return {"ok": True, "accepted_draws_compared": n_log, "draw_lists_bit_identical": True, "max_rel_err": float("%.3g" % worst),
            # max_rel_err is against the exact (fp64) sums the oracle keeps beside its buffers; the next figure is against the
            # oracle's own fp32 buffers and images -- the reference CPU imager's numbers as it would write them -- and the one
            # after it says how far those are themselves from the exact sums (sequential fp32 summation)
            "max_rel_err_vs_fp32_oracle": float("%.3g" % st.get("vs_fp32", float("nan"))),
            "fp32_oracle_own_rounding": float("%.3g" % st.get("fp32_own", float("nan"))),
            "tolerance": 1e-5, "redistributed_visits": int(c.redistributed_visits),
            "sample": "every %d-th row of the timed %dx%d frame (%d visits), HIP path against the oracle" % (row_step, W, H, n),
            "full_size": full_size or
                         "tests/test_gpu_headline.py::test_headline_4k_streamed_vs_oracle (the timed streams, bit-identical "
                         "draw lists, 1e-5 radiance) and ::test_config5_quarter_frame_vs_oracle"}


def power_cap_watts():
    """the package power cap from sysfs (hwmon power1_cap, microwatts); None where it cannot be read.  (No child process: under
    rocprofv3 a `#!/usr/bin/env python3` tool such as rocm-smi would be an exec behind the profiler's preloaded library.)"""
    import glob
    best = None
    for p in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_cap"):
        try:
            with open(p) as f:
                w = int(f.read().strip()) / 1e6
            if w > 0 and (best is None or w > best):
                best = w
        except (OSError, ValueError):
            pass
    return best


def checked(fn, *a, **kw):
    """parity_check's verdict as a JSON object, whatever happens inside it"""
    try:
        return fn(*a, **kw)
    except AssertionError as e:
        return {"ok": False, "error": str(e)[:300]}
    except Exception as e:
        return {"ok": None, "error": "check did not run: %r" % (e,)}


def load_traffic(workload_tag):
    """HBM bytes per scan launch from a separate rocprofv3 --pmc pass (profiles/pmc_scan_latest.json); None unless
    that file was taken on this very workload and kernel."""
    path = os.path.join(ROOT, "profiles", "pmc_scan_latest.json")
    try:
        with open(path) as f:
            d = json.load(f)
        if d.get("workload") == workload_tag:
            return d.get("hbm_bytes_per_launch")
    except Exception:
        pass
    return None


class Bench:
    """One context + two resident visit streams of one frame geometry; run() times alternating passes."""

    def __init__(self, torch, dist, dev, local_rank, world, rank, W, H, Hr, M, lens, samples, aovs, f_hi, bokeh_image,
                 emulate=None, bounds=None, tiled=False, same_frame=False):
        import numpy as np
        from pota_amd import camera, capi, distributed, lens_io
        self.torch, self.dist, self.dev = torch, dist, dev
        self.world, self.rank = world, rank
        self.W, self.H, self.Hr, self.M = W, H, Hr, M
        self.aovs, self.f_hi, self.emulate, self.tiled = aovs, f_hi, emulate, tiled
        self.n_streams = 1 if same_frame else 2
        p = camera.default_params()
        camera.setup_filter(p, W, H, filter_width=1.0, aa_samples=3)
        p, model = camera.setup_po(p, lens, focus_dist=150.0)
        p.samples_override = samples
        if bokeh_image:
            p.bokeh_enable_image = 1
        self.p, self.model = p, model
        self.bokeh_tables = None
        self.table, self.keep = lens_io.make_lens_table(model.spec)
        self.flops = lens_io.newton_iteration_flops(model.spec)
        self.tan_half_fov = float(p.sensor_width) * 0.5 / float(p.focal_length)
        ctx = capi.Context(local_rank)
        ctx.set_params(p)
        ctx.set_lens(self.table)
        # a lens without a kernel built into the library gets one at run time (lentil_lens_jit.h): the timed passes use it
        try:
            ctx.lens_jit_wait(600.0)
        except Exception as e:      # noqa: BLE001  (the table interpreter serves the lens; the line says so)
            sys.stderr.write("bench.py: %r\n" % (e,))
        st, sec = ctx.lens_jit_status()
        self.lens_kernel = {"kind": {0: "built into the library" if ctx.lens_is_compiled() else "table interpreter", 1: "table interpreter (still compiling)",
                                     2: "specialised at run time (hiprtc)", -1: "table interpreter (run-time compilation failed)"}.get(st, str(st)),
                            "compile_seconds": round(sec, 1)}
        if bokeh_image:
            from pota_amd import bokeh
            tex = np.load(os.path.join(ROOT, "tests", "golden", "example_bokeh_kernel_u8.npy")).astype(np.float32) / np.float32(255)
            self.bokeh_tables = bokeh.build_tables(tex)
            ctx.set_bokeh(self.bokeh_tables)
        ctx.alloc_frame(1 + aovs)
        self.ctx = ctx
        self.engine = distributed.HipEngine(ctx, rows=p.yres)
        self.distributed = distributed
        self.native = False          # the exchange inside liblentil_hip.so (main() switches it on after its self-check)
        self.bounds = bounds
        self.band = None
        self.streams = []
        self.set_band()

    def set_band(self):
        from pota_amd import workload
        d = self.distributed
        if self.emulate:
            self.band = d.band_of(self.emulate[1], self.emulate[0], self.H, self.p.yres, self.bounds)
        elif self.tiled:
            self.band = d.band_of(self.rank, self.world, self.H, self.p.yres, self.bounds)
        if self.band is not None:
            self.v_begin = self.band[0] * self.W * self.M
            self.v_end = min(self.band[1], self.H) * self.W * self.M
            self.n_local = self.v_end - self.v_begin
        else:
            self.n_local = workload.frame_visit_count(self.W, self.H, self.M, self.world, self.rank)

    def generate(self, f_hi):
        """(re)creates the resident streams: same geometry, different seeds"""
        from pota_amd import capi, workload
        torch = self.torch
        self.streams = []
        torch.cuda.empty_cache()
        for seed in SEEDS[:self.n_streams]:
            if self.band is not None:
                cols = workload.generate(torch, self.v_begin, self.v_end, self.W, self.H, self.M, seed=seed, f_hi=f_hi,
                                         focus_dist=150.0, tan_half_fov=self.tan_half_fov, n_extra=self.aovs, device=self.dev)
                v, kv = capi.make_visits(cols, visits_per_pixel=self.M, pixels_per_row=self.W, pixel_y0=self.band[0],
                                         ptr=lambda t: t.data_ptr())
            else:
                cols = workload.generate(torch, 0, self.n_local, self.W, self.H, self.M, seed=seed, f_hi=f_hi, focus_dist=150.0,
                                         tan_half_fov=self.tan_half_fov, n_extra=self.aovs, device=self.dev,
                                         row_stride=self.world, row_offset=self.rank)
                v, kv = capi.make_visits(cols, visits_per_pixel=self.M, pixels_per_row=self.W, pixel_y0=self.rank,
                                         pixel_row_stride=self.world, ptr=lambda t: t.data_ptr())
            self.streams.append((cols, v, kv))
        torch.cuda.synchronize()
        self.i_stream = 0

    def step(self):
        cols, v, kv = self.streams[self.i_stream % len(self.streams)]
        self.i_stream += 1
        self.ctx.bind_visits(v, kv)
        if self.native and self.tiled:
            self.distributed.frame_step_bands_native(self.ctx, self.H, self.bounds)
        elif self.native:
            self.distributed.frame_step_native(self.ctx)
        elif self.tiled:
            self.distributed.frame_step_bands(self.engine, self.dist, self.H, self.p.yres, self.bounds)
        else:
            self.distributed.frame_step(self.engine, self.dist)

    def band_accumulators(self):
        """copy of the accumulator rows this rank owns after a step (the whole frame when not tiled)"""
        self.ctx.sync()
        self.ctx.accum_buffer()                      # folds what the scan keeps apart into the block
        self.ctx.sync()
        acc = self.engine.accum
        if self.band is not None and self.tiled:
            per_row = acc.numel() // self.p.yres
            acc = acc[self.band[0] * per_row:self.band[1] * per_row]
        out = acc.clone()
        self.torch.cuda.synchronize()                # the copy ran on torch's stream; the next step's clear is on the library's
        return out

    def check_native_exchange(self):
        """One step through the library's own RCCL exchange and one through torch.distributed (the Python form of the
        same step, which the gloo tests cover), same visit stream: the rank's part of the accumulators must agree to
        fp32 summation order.  Collective; every rank gets the same verdict."""
        torch, dist = self.torch, self.dist
        ok, rel = 1, float("nan")
        try:
            i0 = self.i_stream
            self.native = False
            self.step()
            ref = self.band_accumulators()
            self.i_stream = i0
            self.native = True
            self.step()
            got = self.band_accumulators()
            self.i_stream = i0
            scale = float(ref.abs().max().item())
            rel = float((got - ref).abs().max().item()) / max(scale, 1e-30)
            same_pixels = bool(((got != 0) == (ref != 0)).all().item())
            ok = 1 if (scale > 0 and rel < 1e-4 and same_pixels) else 0
        except Exception as e:
            sys.stderr.write("bench.py rank %d: native exchange self-check raised %r\n" % (self.rank, e))
            ok = 0
        t = torch.tensor([ok], dtype=torch.int64, device=self.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        self.native = bool(int(t.item()))
        return {"ok": self.native, "max_abs_diff_over_max_rank0": rel}

    def run(self, steps, warmup):
        """W warm-up steps, then K timed steps between two synchronisations.  Nothing inside the timed loop waits for the
        device: a step is clear -> pass -> resolve, enqueued; the library looks at a pass's end (did everything fit? is
        more work needed?) when the context is next observed, and a pass whose frame is cleared before that is counted in
        `abandoned_incomplete` if it had needed more work (lentil_hip_pass_totals).  Should any timed pass have been
        abandoned incomplete, the measurement is repeated with every pass waiting for its own end (set_async(0)) and THAT
        is reported -- `timed_loop` in the result says which."""
        r = self._run(steps, warmup)
        if r["abandoned_incomplete"]:
            self.ctx.set_async(False)
            try:
                r2 = self._run(steps, warmup)
            finally:
                self.ctx.set_async(True)
            r2["timed_loop"] = ("every pass waited for its own end: %d of the %d pipelined passes had been cleared away while they "
                                "still needed work" % (r["abandoned_incomplete"], steps))
            return r2
        return r

    def _run(self, steps, warmup):
        torch, dist = self.torch, self.dist
        for _ in range(warmup):
            self.step()
        self.ctx.sync()                                # (observes the warm-up passes: their ends are looked at, untimed)
        self.ctx.pass_totals(reset=True)
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.ctx.sync()                                # the last pass's end, and whatever it left to do: inside the timed region
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if self.world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=self.dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        # per-pass counters and HIP-event times, summed by the library as it looked at each pass's end
        k = self.ctx.pass_totals(reset=True)
        if int(k.passes) != steps:
            raise SystemExit("bench.py: %d timed steps but the library accounted for %d passes" % (steps, int(k.passes)))
        if k.worklist_overflow:
            raise SystemExit("bench.py: the device dropped work (worklist_overflow = %d): results incomplete" % k.worklist_overflow)
        acc = {"scan": float(k.scan_ms), "draw": float(k.draw_ms), "resolve": float(k.resolve_ms), "streamed": int(k.streamed),
               "blind_chunks": int(k.blind_chunks), "redone": int(k.fallback_chunks), "iters": int(k.newton_iterations),
               "attempted": int(k.attempted_draws), "accepted": int(k.accepted_draws), "redistributed": int(k.redistributed_visits),
               "scan_launches": int(k.scan_launches), "lane_rounds": int(k.lane_rounds), "tries": int(k.tries), "slow": int(k.slow_solves),
               "rounds_max": int(k.rounds_max), "deferred": int(k.deferred), "abandoned": int(k.abandoned),
               "abandoned_incomplete": int(k.abandoned_incomplete),
               "timed_loop": "pipelined: no host wait inside the loop, every pass's end looked at by the library afterwards"
                             if int(k.deferred) else "every pass waited for its own end"}
        if int(k.fallback_chunks):
            # why (lentil_hip_last_redo_note): a pass redone costs time, never results -- but a bench line should say so
            acc["redo_notes"] = [self.ctx.last_redo_note()]
        acc["dt"] = dt
        acc["steps"] = steps
        return acc

    def solve_block(self, r):
        """the draw kernels against the fp64 vector peak; the time is everything but the resolve (in a streamed pass
        the solves run beside the scan)"""
        mul, add, other = self.flops
        per_iter = mul + add + other
        secs = (r["scan"] + r["draw"]) * 1e-3
        tf = r["iters"] * per_iter / secs / 1e12 if secs > 0 else 0.0
        return {
            "kernel": "solve_po_kernel + solve_slow_kernel", "bound": "fp64 valu",
            "lane_iterations_per_step": r["iters"] // max(1, r["steps"]),
            "solves_per_step": r["tries"] // max(1, r["steps"]), "parked_solves_per_step": r["slow"] // max(1, r["steps"]),
            "lane_utilisation": round(r["iters"] / r["lane_rounds"], 4) if r["lane_rounds"] else None,
            "flops_per_lane_iteration": {"mul": mul, "add": add, "transforms_and_inverses": other},
            "ms_per_step_scan_plus_draw": round(secs * 1e3 / max(1, r["steps"]), 4),
            "achieved": round(tf, 3), "unit": "TFLOP/s", "peak_fma": FP64_VECTOR_PEAK_TFLOPS,
            "frac_of_fma_peak": round(tf / FP64_VECTOR_PEAK_TFLOPS, 4),
            "frac_of_non_fma_ceiling": round(tf / (FP64_VECTOR_PEAK_TFLOPS / 2), 4),
        }

    def close(self):
        self.streams = []
        self.ctx.close()
        self.torch.cuda.empty_cache()


def summarize(b, r, n_total, bytes_per_visit):
    ms = r["dt"] / r["steps"] * 1e3
    launches = max(1, r["scan_launches"] // r["steps"])
    launch_ms = r["scan"] / r["steps"] / launches
    launch_bytes = b.n_local * bytes_per_visit / launches
    achieved = launch_bytes / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
    return ms, launches, launch_ms, launch_bytes, achieved


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child (the launcher the driver itself uses),
    BEFORE this process has imported torch or touched the GPU, pass its output on and leave with its exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("bench.py: starting %d ranks: %s\n" % (args.gpus, " ".join(cmd)))
    p = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    js = [l for l in lines if l.lstrip().startswith("{") and '"metric"' in l]
    for l in lines:
        if not js or l is not js[-1]:
            sys.stderr.write(l + "\n")          # (launcher / RCCL chatter)
    if js:
        print(js[-1], flush=True)
    elif p.returncode == 0:
        sys.stderr.write("bench.py: the ranks printed no result line\n")
        sys.exit(1)
    sys.exit(p.returncode)


def rank_details(b, dist, torch, dev, steps=3):
    """Per-rank pass / exchange times (host clock around the two halves of the step, a stream sync after each; untimed
    extra steps) and what the rank sent and received in its last exchange; gathered to every rank."""
    mine = {"rank": b.rank, "pass_ms": None, "exchange_ms": None, "sent_bytes": None, "received_bytes": None, "band_rows": list(b.band) if b.band else None}
    if b.native and b.tiled:
        tp = te = 0.0
        for _ in range(steps):
            cols, v, kv = b.streams[b.i_stream % len(b.streams)]
            b.i_stream += 1
            b.ctx.bind_visits(v, kv)
            b.ctx.set_closest_exchange(False)
            dist.barrier()
            t0 = time.perf_counter()
            b.ctx.clear_frame(); b.ctx.redistribute(); b.ctx.sync()
            t1 = time.perf_counter()
            b.ctx.exchange_bands(b.H, b.bounds, sparse=b.distributed.SPARSE_EXCHANGE); b.ctx.sync()
            t2 = time.perf_counter()
            tp += t1 - t0; te += t2 - t1
        sent, recv = b.ctx.exchange_stats()
        mine.update(pass_ms=round(tp / steps * 1e3, 3), exchange_ms=round(te / steps * 1e3, 3), sent_bytes=sent, received_bytes=recv)
    allr = [None] * b.world
    dist.all_gather_object(allr, mine)
    return allr


def scan_kernel_name(aovs):
    """The scan kernel the library picks for a uniform stream of whole pixels (plan_scan in lentil_hip.hip)."""
    dma = os.environ.get("LENTIL_SCAN_DMA", "1") != "0"
    if aovs == 0:
        if dma and os.environ.get("LENTIL_SCAN_DMA2", "1") != "0":
            return "scan_dma2_kernel"       # (round 4: tiles pipelined into one another, one block per CU)
        return "scan_dma_kernel" if dma else "scan_uniform_kernel"
    return "scan_dma_multi_kernel" if (dma and os.environ.get("LENTIL_SCAN_DMA_MULTI", "1") != "0") else "scan_uniform_multi_kernel"


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.emulate:
        spawn_ranks(args)
    import torch
    import torch.distributed as dist
    from pota_amd import capi, workload

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    emulate = tuple(int(x) for x in args.emulate.split(",")) if args.emulate else None
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and not emulate:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    backend = os.environ.get("LENTIL_DIST_BACKEND", "nccl")     # "gloo": several ranks on one GPU (development aid)
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    force_dist = os.environ.get("LENTIL_FORCE_DIST") == "1"     # exercise the RCCL plumbing on one GPU
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    dev = torch.device("cuda", local_rank)
    ranks_joined = 1
    if world > 1:
        # self-check of the launch: every rank the driver asked for has joined
        t = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(t)
        ranks_joined = int(t.item())
        if ranks_joined != world or world != args.gpus:
            raise SystemExit("bench.py: %d ranks joined, --gpus %d, WORLD_SIZE %d" % (ranks_joined, args.gpus, world))

    M = args.visits_per_pixel
    # Weak scaling: N GPUs render the same camera at N times the pixels (same 16:9 frame, sqrt(N) finer in both
    # directions), so every rank's band has the per-GPU pixel count, highlight count and field-angle mix of the
    # single-GPU frame.  (Stacking N frames on top of each other would push the outer bands far off axis, where the
    # lens passes nothing and every visit burns its 5 x samples attempts.)
    frame_world = emulate[0] if emulate else world
    strong = args.scaling == "strong"          # (also under --emulate: rank r's band of the strong-scaled frame)
    if frame_world == 1:
        W, Hr = args.width, args.height
        H = Hr
    elif strong:
        # the metric's wording: the 4K frame at N GPUs -- the same frame, tiled
        W, H = args.width, args.height
        Hr = (H + frame_world - 1) // frame_world
    else:
        W = int(round(args.width * math.sqrt(frame_world)))
        Hr = int(round(args.height / math.sqrt(frame_world)))
        H = Hr * frame_world
    tiled = (world > 1 or force_dist) and os.environ.get("LENTIL_PARTITION", "bands") != "interleaved" and not emulate
    bounds = [int(x) for x in args.bounds.split(",")] if args.bounds else None

    b = Bench(torch, dist, dev, local_rank, world, rank, W, H, Hr, M, args.lens, args.samples, args.aovs, args.f_hi,
              args.bokeh_image, emulate=emulate, bounds=bounds, tiled=tiled, same_frame=args.same_frame)
    bytes_per_visit = 80 + 16 * args.aovs
    b.generate(args.f_hi)
    exchange = None
    if (world > 1 or force_dist) and not emulate:
        exchange = {"impl": "torch.distributed (pota_amd/distributed.py)", "self_check": None}
        if backend == "nccl" and os.environ.get("LENTIL_EXCHANGE_IMPL", "native") == "native":
            try:
                b.distributed.native_comm_init(b.ctx, dist)
                have = 1
            except Exception as e:
                sys.stderr.write("bench.py rank %d: no native communicator: %r\n" % (rank, e))
                have = 0
            t = torch.tensor([have], dtype=torch.int64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            if int(t.item()):
                exchange["self_check"] = b.check_native_exchange()
                if b.native:
                    exchange["impl"] = ("liblentil_hip.so: %s (RCCL bound by the library)"
                                        % ("lentil_hip_exchange_bands" if tiled else "lentil_hip_allreduce"))
    if tiled and (world > 1 or force_dist) and not args.bounds and os.environ.get("LENTIL_REBALANCE", "1") != "0":
        # calibration (untimed set-up, before the warm-up steps): three passes, the first with even bands, each followed
        # by an all-gather of the ranks' pass times (scan + draws, HIP events) and a re-cut of the bands
        # (the first cut from the cost model of the frame's edges, not the even one: LENTIL_REBALANCE_START=even restores that)
        b.bounds = (b.distributed.even_bounds(world, H) if os.environ.get("LENTIL_REBALANCE_START", "model") == "even"
                    else b.distributed.modelled_bounds(world, H))
        if b.bounds != b.distributed.even_bounds(world, H):
            b.set_band()
            b.generate(args.f_hi)
        for it in range(3):
            b.step()
            a_, b_, c_ = b.ctx.last_timing()
            mine = torch.tensor([a_ + b_], dtype=torch.float64, device=dev)
            allt = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allt, mine)
            times = torch.stack(allt).flatten().tolist()
            new = b.distributed.rebalance(b.bounds, times, damping=(1.0, 0.8, 0.6)[it])
            if new != b.bounds:
                b.bounds = new
                b.set_band()
                b.generate(args.f_hi)
    else:
        # set-up: the first pass of a context sizes its draw buffers from the scan's counters (host round trips,
        # allocations); two passes settle that, whatever --warmup says
        for _ in range(2):
            b.step()
        torch.cuda.synchronize()
    # what this box delivers, before and after the timed steps (rank 0's GPU; the probe's kernels run ~10 ms on the library's stream)
    box = None
    if rank == 0:
        box = checked(b.ctx.box_probe)
    r = b.run(args.steps, args.warmup)
    if rank == 0 and isinstance(box, dict) and "error" not in box:
        after = checked(b.ctx.box_probe)
        if isinstance(after, dict) and "error" not in after:
            box["after_the_timed_steps"] = {k: after[k] for k in ("fp64_mul_add_tflops", "shader_clock_mhz_under_fp64", "copy_gbs", "read_gbs")}
        box["power_cap_w"] = power_cap_watts()
        box["note"] = ("measured in this run by lentil_hip_box_probe: dependent fp64 multiply/add chains at three waves per SIMD (the solves' "
                       "arithmetic, no FMA; ceiling = CUs x 4 SIMDs x 16 lanes x clock), clock64() against the 100 MHz counter under that load, "
                       "a 512 MiB float4 copy (bytes read + written) and read-only stream; for putting timings of different boxes side by side")
    # the clock those steps ran at: a short loop of its own with a one-wave sampler beside it (tools/clock_trace.py; never the
    # timed loop) -- box.shader_clock_mhz_under_fp64 is what the chip settles at after 15 ms of steady fp64 load, a 2-ms pass gets less
    clock_in_pass = None
    if rank == 0 and world == 1 and not args.no_clock_trace:
        try:
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
            import clock_trace
            clock_in_pass = clock_trace.clock_in_pass(b, r["dt"] / r["steps"] * 1e3)
        except Exception as e:      # a measuring aid: the line is reported without it
            clock_in_pass = {"error": repr(e)}
    # ... and the same workload again after 350 untimed steps (0.7 s).  The chip's power management moves between regimes over tenths
    # of a second to seconds: on most boxes an idle chip's first 0.4 s of passes run at ~2.0 GHz in the solve phase (2.0 ms a step)
    # and then settle at ~2.25 GHz (1.86 ms); a chip that has been loaded for some seconds may sit at 2.0 GHz for good
    # (tools/sustained.py, profiles/r06_sustained.txt).  W warm-up steps and K timed ones (`value`: 3 + 20 as the driver runs it, 46 ms)
    # see whichever regime the chip is in; this second figure, timed the same way, is reported beside it in roofline.sustained --
    # never as `value` -- with the clock in the pass for both.
    sustained = None
    if rank == 0 and world == 1 and not args.no_sustained and not emulate:
        try:
            for _ in range(350):
                b.step()
            rs = b.run(200, 0)
            ms_s = rs["dt"] / rs["steps"] * 1e3
            sustained = {"ms_per_step": round(ms_s, 4), "steps": 200, "untimed_steps_before": 350,
                         "kernels_ms": {"scan": round(rs["scan"] / rs["steps"], 4), "draw": round(rs["draw"] / rs["steps"], 4)},
                         "passes": {"streamed": rs["streamed"], "redone": rs["redone"], "rounds_max": rs.get("rounds_max")}}
            if isinstance(clock_in_pass, dict) and "error" not in clock_in_pass:
                import clock_trace
                c2 = clock_trace.clock_in_pass(b, ms_s)
                sustained["clock_in_pass"] = {k: c2[k] for k in ("mean_mhz", "middle_half_of_a_step_mhz")}
        except Exception as e:      # a second figure: the line is reported without it
            sustained = {"error": repr(e)}
    n_total = workload.frame_visit_count(W, H, M) if not emulate else b.n_local        # all ranks
    value = n_total * r["steps"] / r["dt"] / 1e6
    ms_per_step, launches, launch_ms, launch_bytes, achieved = summarize(b, r, n_total, bytes_per_visit)
    scan_kernel = scan_kernel_name(args.aovs)
    workload_tag = "%s %s %dx%d M=%d samples=%d aovs=%d f_hi=%.3g" % (scan_kernel, args.lens, W, Hr, M, args.samples, 1 + args.aovs, args.f_hi)
    # bytes the scan requests per visit: four of the five base columns (raydir_time only for visits at infinite
    # depth) + the extra AOV columns, and per pixel one record stored (and, in the register-staged kernels, read first)
    rec_bytes = 4 * (((4 * (1 + args.aovs) + 1) + 7) // 8 * 8)
    moved = 64 + 16 * args.aovs + (rec_bytes if scan_kernel.startswith("scan_dma") else 2 * rec_bytes) / M
    steps = r["steps"]
    out = {
        "metric": "bidir redistribution Msamples/s at 4K, double-gauss 50mm",
        "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak" if (args.scaling == "weak" and not emulate) else "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": "polynomial-optics %s (self-fitted table), frame %dx%d (%s), "
                        "~%d rows per GPU, %d visits/pixel, %d redistribution draws per redistributed visit, %d AOV(s), "
                        "highlight fraction f_hi=%.3g; %s"
                        % (args.lens, W, H,
                           ("the %dx%d frame tiled over %d GPU(s)" % (args.width, args.height, world)) if (strong or world == 1)
                           else ("%d x the pixels of %dx%d, same camera" % (frame_world, args.width, args.height)),
                           Hr, M, args.samples, 1 + args.aovs, args.f_hi,
                           "ONE visit stream replayed" if args.same_frame else
                           "timed steps alternate between two visit streams (seeds %#x / %#x)" % SEEDS),
            "visits_per_gpu": b.n_local, "bytes_per_visit": bytes_per_visit,
            "redistributed_visits_per_step_rank0": r["redistributed"] // steps,
            "attempted_draws_per_step_rank0": r["attempted"] // steps, "accepted_draws_per_step_rank0": r["accepted"] // steps,
            "parallelism": ("single GPU" if world == 1 else
                            "%d row bands%s, rows touched outside a band sent to its owner (p2p), tiled output"
                            % (world, (" at rows %s (balanced by pass time)" % b.bounds) if b.bounds else "") if tiled else
                            "rows%%%d + allreduce" % world),
        },
        "box": box, "streams_concurrent": b.ctx.streams_concurrent(),
        "exchange": exchange, "ranks_joined": ranks_joined,
        "passes": {"timed": steps, "streamed": r["streamed"], "chunks_enqueued_blind": r["blind_chunks"],
                   "chunks_redone_after_a_short_estimate": r["redone"], "solve_accept_rounds_max": r.get("rounds_max"),
                   "redo_notes": r.get("redo_notes", []),
                   # the whole process so far (lentil_hip_process_stats): streamed passes begun, those whose resident waves hit the
                   # stuck time-out (each wiped and run again: right, and late), how many of those a test asked for, passes redone
                   "process": dict(zip(("streamed", "stuck", "stuck_asked_for", "redone"), capi.process_stats())),
                   # the asynchronous end of a pass (include/lentil_hip.h, lentil_hip_set_async): passes that returned before their
                   # end was known, those whose frame was cleared before anybody observed it, and how many of those had still needed work
                   "timed_loop": r.get("timed_loop"), "ends_deferred": r.get("deferred"), "abandoned": r.get("abandoned"),
                   "abandoned_incomplete": r.get("abandoned_incomplete"),
                   # first batches sized from the lens and the frame (lentil_hip_batch_model_stats, whole life of the context):
                   # passes that ran with no second round of solves in flight, how many of those needed one after all
                   "first_batch_model": dict(zip(("calibrations", "lean_passes", "lean_passes_lost", "margin_sixteenths"),
                                                 b.ctx.batch_model_stats()))},
        "kernels_ms": {"scan": round(r["scan"] / steps, 4), "draw": round(r["draw"] / steps, 4), "resolve": round(r["resolve"] / steps, 4)},
        "roofline": {
            "kernel": scan_kernel, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": load_traffic(workload_tag),
            "traffic_source": "profiles/pmc_scan_latest.json (separate rocprofv3 --pmc passes; null unless taken on this workload)",
            "launches_per_step": launches, "algorithmic_bytes_per_launch": round(launch_bytes),
            "avg_launch_ms": round(launch_ms, 4),
            "algorithmic_bytes_per_visit": bytes_per_visit, "bytes_moved_per_visit": round(moved, 2),
            "whole_step_frac": round(b.n_local * bytes_per_visit / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
        },
        "solve_fp64": {"scan_dominated": b.solve_block(r)},
    }

    if world > 1 and not emulate:
        try:
            out["ranks"] = rank_details(b, dist, torch, dev)
        except Exception as e:      # the headline number must still be reported
            out["ranks"] = {"error": repr(e)}
        if not args.no_config5:
            # BASELINE config 5: 7680x4320, 2048 draws (above the reference's clamp of 2000: samples_override), the frame
            # tiled across the N GPUs, cross-tile splats exchanged over xGMI
            name = "config5_double_gauss_7680x4320_2048_draws_tiled_%d_gpus" % world
            try:
                W5, H5 = 7680, 4320
                c5 = Bench(torch, dist, dev, local_rank, world, rank, W5, H5, (H5 + world - 1) // world, M, args.lens, 2048, 0,
                           args.f_hi, False, tiled=tiled)
                c5.generate(args.f_hi)
                c5.native = False
                if b.native:
                    c5.distributed.native_comm_init(c5.ctx, dist)
                    c5.native = True
                for _ in range(2):
                    c5.step()
                r5 = c5.run(4, 2)
                n5 = workload.frame_visit_count(W5, H5, M)
                out.setdefault("configs", {})[name] = {
                    "value": round(n5 * r5["steps"] / r5["dt"] / 1e6, 3), "unit": "Msamples/s", "ms_per_step": round(r5["dt"] / r5["steps"] * 1e3, 4),
                    "kernels_ms_rank0": {"scan": round(r5["scan"] / 4, 4), "draw": round(r5["draw"] / 4, 4), "resolve": round(r5["resolve"] / 4, 4)},
                    "attempted_draws_per_step_rank0": r5["attempted"] // 4, "steps": 4,
                    "whole_step_frac_of_hbm_peak_per_gpu": round(n5 * 80 / world / (r5["dt"] / r5["steps"]) / 1e9 / HBM_PEAK_GBS, 4),
                    "exchange": "native" if c5.native else "torch.distributed", "ranks": rank_details(c5, dist, torch, dev, steps=2)}
                c5.close()
            except Exception as e:
                out.setdefault("configs", {})[name] = {"value": None, "error": repr(e)}

    if rank == 0 and world == 1 and not emulate and not args.no_configs and os.environ.get("GPU_MAX_HW_QUEUES") != "4":
        # The same headline pass in a process whose runtime keeps its default of four hardware queues (a renderer that loads the
        # plugin sets nothing): a child process, because the runtime reads the variable when it initialises.  lentil_hip_create
        # probes whether the pass's streams run side by side there; where they do not, the pass takes its chunked form.
        try:
            import subprocess
            env = dict(os.environ, GPU_MAX_HW_QUEUES="4")
            cmd = [sys.executable, os.path.abspath(__file__), "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-second-regime",
                   "--no-configs", "--no-pcie", "--no-parity-check", "--no-scan-alone", "--width", str(args.width), "--height", str(args.height),
                   "--samples", str(args.samples), "--lens", args.lens, "--aovs", str(args.aovs)]
            line = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600).stdout.strip().splitlines()[-1]
            d4 = json.loads(line)
            out["runtime_default_hw_queues"] = {
                "GPU_MAX_HW_QUEUES": 4, "ms_per_step": d4["ms_per_step"], "value": d4["value"], "passes": d4["passes"],
                "streams_concurrent": d4.get("streams_concurrent"),
                "note": "the headline workload in a child process with the runtime's default of four hardware queues (INTEGRATION.md section 3a)"}
        except Exception as e:      # noqa: BLE001
            out["runtime_default_hw_queues"] = {"ms_per_step": None, "error": repr(e)}

    if not args.no_second_regime and world == 1:
        f2 = 1.6e-3
        b.generate(f2)
        st = max(2, min(args.steps, 2))
        r2 = b.run(st, 2)
        out["regimes"] = {"highlight_heavy": {
            "f_hi": f2, "value": round(n_total * st / r2["dt"] / 1e6, 3), "unit": "Msamples/s", "steps": st,
            "ms_per_step": round(r2["dt"] / st * 1e3, 3),
            "kernels_ms": {"scan": round(r2["scan"] / st, 4), "draw": round(r2["draw"] / st, 4), "resolve": round(r2["resolve"] / st, 4)},
            "attempted_draws_per_step": r2["attempted"] // st,
            "passes": {"streamed": r2["streamed"], "chunks_redone_after_a_short_estimate": r2["redone"]}}}
        out["solve_fp64"]["highlight_heavy"] = b.solve_block(r2)

    if rank == 0 and world == 1 and not args.no_pcie and not emulate:
        try:
            b.generate(args.f_hi)
            out["pcie_inclusive"] = pcie_inclusive(b)
        except Exception as e:      # the GPU number must still be reported
            out["pcie_inclusive"] = {"value": None, "error": repr(e)}
    cpu_args = (args, b.p, b.table, M, b.tan_half_fov)
    cpu_bokeh = b.bokeh_tables
    b.close()

    def scan_alone(Wc, Hc, lens, samples, aovs, bokeh, n_c, bpv):
        """The same scan kernel with the chip to itself: the streamed pass makes it share its CUs with the solve waves that
        take its output as it comes (roofline.frac is that launch).  Here: the chunked form of the pass, one chunk -- the
        scan runs to its end before the first solve kernel starts; same streams, same kernel, measured the same way."""
        saved = {k: os.environ.get(k) for k in ("LENTIL_STREAM", "LENTIL_CHUNKS")}
        os.environ["LENTIL_STREAM"] = "0"; os.environ["LENTIL_CHUNKS"] = "1"
        try:
            c = Bench(torch, dist, dev, local_rank, 1, 0, Wc, Hc, Hc, M, lens, samples, aovs, args.f_hi, bokeh)
            c.generate(args.f_hi)
            for _ in range(2):
                c.step()
            rc = c.run(4, 1)
            _, l_a, lms_a, lb_a, ach_a = summarize(c, rc, n_c, bpv)
            c.close()
            return {"avg_launch_ms": round(lms_a, 4), "achieved": round(ach_a, 1), "frac": round(ach_a / HBM_PEAK_GBS, 4),
                    "launches_per_step": l_a, "how": "LENTIL_STREAM=0 LENTIL_CHUNKS=1: the scan ends before the solves start"}
        except Exception as e:
            return {"frac": None, "error": repr(e)}
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v

    if rank == 0 and world == 1 and not emulate and not args.no_scan_alone:
        out["roofline"]["alone"] = scan_alone(W, H, args.lens, args.samples, args.aovs, args.bokeh_image, n_total, bytes_per_visit)

    if not args.no_configs and world == 1 and not emulate:
        # BASELINE.json's other single-GPU configurations, measured the same way (alternating streams)
        cfgs = {}
        for name, kw in (
                ("config2_double_gauss_1920x1080_256_draws", dict(W=1920, H=1080, lens="double_gauss_50mm", samples=256, aovs=0, bokeh=False)),
                ("config3_double_gauss_bokeh_image_3840x2160_512_draws", dict(W=3840, H=2160, lens="double_gauss_50mm", samples=512, aovs=0, bokeh=True)),
                ("config4_petzval_3840x2160_1024_draws_9_aovs", dict(W=3840, H=2160, lens="petzval_58mm", samples=1024, aovs=8, bokeh=False)),
                # config 4's other half, "anamorphic": a table with a cylindrical front element and NO kernel built into the library --
                # its solve kernels are emitted and compiled at run time (lens_kernel says which kernel the timed passes ran)
                ("config4_anamorphic_petzval_3840x2160_1024_draws_9_aovs", dict(W=3840, H=2160, lens="anamorphic_petzval_58mm", samples=1024, aovs=8, bokeh=False))):
            try:
                c = Bench(torch, dist, dev, local_rank, 1, 0, kw["W"], kw["H"], kw["H"], M, kw["lens"], kw["samples"], kw["aovs"],
                          args.f_hi, kw["bokeh"])
                c.generate(args.f_hi)
                for _ in range(2):
                    c.step()
                rc = c.run(4, 2)
                bpv = 80 + 16 * kw["aovs"]
                n_c = workload.frame_visit_count(kw["W"], kw["H"], M)
                ms_c, l_c, lms_c, lb_c, ach_c = summarize(c, rc, n_c, bpv)
                cfgs[name] = {"value": round(n_c * rc["steps"] / rc["dt"] / 1e6, 3), "unit": "Msamples/s", "ms_per_step": round(ms_c, 4),
                              "kernels_ms": {"scan": round(rc["scan"] / 4, 4), "draw": round(rc["draw"] / 4, 4), "resolve": round(rc["resolve"] / 4, 4)},
                              "scan_kernel": scan_kernel_name(kw["aovs"]), "bytes_per_visit": bpv, "scan_frac_of_hbm_peak": round(ach_c / HBM_PEAK_GBS, 4),
                              "whole_step_frac_of_hbm_peak": round(n_c * bpv / (ms_c * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                              "attempted_draws_per_step": rc["attempted"] // 4, "passes_streamed": rc["streamed"], "steps": 4,
                              "lens_kernel": c.lens_kernel}
                pc = (c.p, c.table, c.tan_half_fov, c.bokeh_tables)
                c.close()
                if not args.no_parity_check:
                    test = {"config2": "test_config2_1080p_256_draws_vs_oracle", "config3": "test_config3_4k_512_draws_aperture_image_vs_oracle",
                            "config4": "test_config4_like_4k_nine_gaussian_aovs_streamed_vs_oracle (and, with two closest-filtered "
                                       "AOVs, ::test_config4_4k_petzval_two_closest_aovs_vs_oracle)"}[name[:7]]
                    if "anamorphic" in name:
                        test = "test_config4_anamorphic_4k_nine_aovs_runtime_kernel_vs_oracle"
                    cfgs[name]["parity_checked"] = checked(
                        parity_check, kw["W"], kw["H"], M, kw["samples"], kw["aovs"], args.f_hi, pc[0], pc[1], pc[2], local_rank, torch,
                        bokeh_tables=pc[3], budget_s=5.0,
                        full_size="tests/test_gpu_headline.py::%s: the whole frame, every pass of the bench's order" % test)
                if not args.no_scan_alone:
                    cfgs[name]["scan_alone"] = scan_alone(kw["W"], kw["H"], kw["lens"], kw["samples"], kw["aovs"], kw["bokeh"], n_c, bpv)
            except Exception as e:      # the headline number must still be reported
                cfgs[name] = {"value": None, "error": repr(e)}
        out["configs"] = cfgs

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(*cpu_args)
        except Exception as e:      # the GPU number must still be reported
            out["cpu_baseline"] = {"value": None, "unit": "Msamples/s", "cores": 0, "kind": "port",
                                   "sample": "failed: %r" % (e,)}
    if rank == 0 and world == 1 and not args.no_parity_check:
        out["parity_checked"] = checked(parity_check, args.width, args.height, M, args.samples, args.aovs, args.f_hi, cpu_args[1],
                                        cpu_args[2], cpu_args[4], local_rank, torch, bokeh_tables=cpu_bokeh)
    from pota_amd import distributed
    if distributed.PHASE_SECONDS:
        sys.stderr.write("[band timing, ms per step incl. warm-up steps] %s\n" % {k: round(v * 1e3 / (args.steps + args.warmup), 3)
                                                                 for k, v in distributed.PHASE_SECONDS.items()})
    used_rccl = dist.is_initialized()
    if used_rccl:
        dist.destroy_process_group()
    if rank == 0:
        # What explains the number rides INSIDE `roofline` (a reader that keeps that object whole and only the names of the other
        # keys still has it): the box probe -- fp64 chain rate, clock, copy / read bandwidth, queues -- and the pass's kernel times;
        # and `roofline` / `cpu_baseline` are the LAST keys of the line, so that a tail of the output holds them.
        out["roofline"]["box"] = out.get("box")
        out["roofline"]["clock_in_pass"] = clock_in_pass
        if isinstance(sustained, dict) and "ms_per_step" in sustained:
            sustained["value"] = round(n_total / (sustained["ms_per_step"] * 1e-3) / 1e6, 3)
            sustained["unit"] = "Msamples/s"
            sustained["whole_step_frac"] = round(b.n_local * bytes_per_visit / (sustained["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
            sustained["note"] = ("the same workload after 350 more untimed steps (0.7 s), timed the same way: the chip's power management moves between "
                                 "a ~2.0 GHz and a ~2.25 GHz regime of the solve phase over tenths of a second to seconds (1.86-2.00 ms a step; "
                                 "profiles/r06_sustained.txt, DESIGN.md section 4.2c); compare the two clock_in_pass blocks")
        out["roofline"]["sustained"] = sustained
        out["roofline"]["kernels_ms"] = out.get("kernels_ms")
        out["roofline"]["ms_per_step"] = out.get("ms_per_step")
        out["roofline"]["timed_loop"] = (out.get("passes") or {}).get("timed_loop")
        for key in ("parity_checked", "roofline", "cpu_baseline"):
            if key in out:
                out[key] = out.pop(key)
        # the single JSON line is the last thing on stdout (RCCL prints a banner of its own)
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    # RCCL prints a three-line banner to stdout from a library destructor; leave without running those so
    # that the JSON line stays the last line of output on every rank.  Single-process runs exit normally
    # (rocprofv3 writes its output files from an exit handler).
    sys.stdout.flush()
    sys.stderr.flush()
    if used_rccl and os.environ.get("LENTIL_BENCH_NO_EXIT") != "1":     # (set it under rocprofv3 with LENTIL_FORCE_DIST)
        os._exit(0)


if __name__ == "__main__":
    main()
