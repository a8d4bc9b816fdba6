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
highlight-heavy regime 1.6e-3 is reported beside it in "regimes").  N>1 keeps the per-GPU work fixed
(weak scaling): the same camera at N times the pixels (16:9, sqrt(N) finer both ways), rank r owns a band
of consecutive rows -- its visits and its tile of the output -- and sends the rows its draws touched
outside the band to their owners (pota_amd/distributed.py::frame_step_bands).
LENTIL_PARTITION=interleaved selects rows r mod N with one sum all-reduce of the whole frame instead.

Set-up before the W warm-up steps (untimed, like generating the visits): two passes that let the context size its
draw buffers (the first pass of a context waits for each chunk's scan and allocates); for N > 1 three calibration
passes that re-cut the row bands from the ranks' pass times.

The printed JSON line also carries
  roofline     -- the HBM-bound scan kernel: algorithmic bytes (visits x 80 B) / its mean duration,
                  measured with HIP events on the library's stream, against 8 TB/s
  cpu_baseline -- the oracle (a port of the reference CPU path) timed on this box's host cores on a
                  bounded row sample of the same workload (rank 0, N=1 only).
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

# A pass uses three HIP streams (scans, one per chunk of draws); torch and RCCL bring theirs.  With the runtime's
# default of 4 hardware queues per process two of them end up sharing one in the multi-GPU runs, and the first
# chunk's solves then queue behind the second scan instead of running beside it (3.34 -> 3.05 ms per step).
# Read when the HIP runtime initialises, so: before torch is imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# dmabuf IPC for RCCL between the ranks of one node (the host driver here supports nothing else); also read early
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_VECTOR_PEAK_TFLOPS = 78.6


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--visits-per-pixel", type=int, default=9)
    ap.add_argument("--samples", type=int, default=1024)
    ap.add_argument("--f-hi", type=float, default=2.0 ** -16)
    ap.add_argument("--aovs", type=int, default=0, help="extra (non-beauty) AOVs")
    ap.add_argument("--lens", default="double_gauss_50mm")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-second-regime", action="store_true")
    ap.add_argument("--cpu-row-step", type=int, default=0, help="CPU baseline samples every n-th row (0 = auto)")
    ap.add_argument("--bokeh-image", action="store_true", help="aperture draws from the reference's example bokeh image "
                                                               "(tests/golden/example_bokeh_kernel_u8.npy; BASELINE config 3)")
    ap.add_argument("--bounds", default="", help="band boundaries (N+1 visit rows, comma separated) instead of the even split")
    ap.add_argument("--emulate", default="", help="development aid: 'N,r' runs rank r's band of the N-GPU frame in one "
                                                   "process (no exchange): per-band cost of the weak-scaling workload")
    return ap.parse_args()


def cpu_baseline(args, p, table, M, tan_half_fov):
    """Oracle ("port" of the reference CPU path) on a bounded sample: every row_step-th image row of
    the same frame, threaded over rows with per-thread private accumulators merged at the end."""
    import ctypes as C
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from pota_amd import capi, workload

    lib = oracle_lib.load()
    W, H = args.width, args.height
    threads = max(1, min(os.cpu_count() or 1, 32))
    # bound the CPU work to roughly 10-30 s: ~0.1 us per scanned visit, ~35 us per draw attempt per core
    est_full = W * H * M * (0.1e-6 + args.f_hi * args.samples * 35e-6 * 1.1)
    row_step = args.cpu_row_step or max(1, int(round(est_full / (20.0 * threads))))
    rows = list(range(0, H, row_step))
    n = len(rows) * W * M
    cols = workload.generate(np, 0, n, W, H, M, f_hi=args.f_hi, focus_dist=150.0, tan_half_fov=tan_half_fov,
                             n_extra=args.aovs, row_stride=row_step, row_offset=0)
    visits, keep = capi.make_visits(cols, visits_per_pixel=M, pixels_per_row=W, pixel_row_stride=row_step)
    lens = lib.orc_lens_create(C.byref(table))
    frames = [oracle_lib.Frame(lib, p, n_aovs=1 + args.aovs, shadow=False) for _ in range(threads)]
    # contiguous row blocks per thread
    bounds = [int(round(i * len(rows) / threads)) * W * M for i in range(threads + 1)]

    def work(i):
        frames[i].run(lens, None, visits, bounds[i], bounds[i + 1])

    t0 = time.perf_counter()
    ts = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for f in frames[1:]:
        lib.orc_frame_merge(frames[0].h, f.h)
    dt = time.perf_counter() - t0
    c = frames[0].counters()
    lib.orc_lens_destroy(lens)
    for f in frames:
        f.close()
    return {
        "value": round(n / dt / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
        "sample": "every %d-th row of the same %dx%d frame: %d visits, %d redistributed, %d draw attempts, %.1f s"
                  % (row_step, W, H, n, c.redistributed_visits, c.attempted_draws, dt),
    }


def load_traffic(workload_tag):
    """HBM bytes per scan launch from a separate rocprofv3 --pmc pass (profiles/pmc_scan_latest.json)."""
    path = os.path.join(ROOT, "profiles", "pmc_scan_latest.json")
    try:
        with open(path) as f:
            d = json.load(f)
        if d.get("workload") == workload_tag:
            return d.get("hbm_bytes_per_launch")
    except Exception:
        pass
    return None


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist
    from pota_amd import camera, capi, distributed, lens_io, workload

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    emulate = None
    if args.emulate:
        emulate = tuple(int(x) for x in args.emulate.split(","))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks" % (args.gpus, args.gpus))
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

    M = args.visits_per_pixel
    # Weak scaling: N GPUs render the same camera at N times the pixels (same 16:9 frame, sqrt(N) finer in both
    # directions), so every rank's band has the per-GPU pixel count, highlight count and field-angle mix of the
    # single-GPU frame.  (Stacking N frames on top of each other would push the outer bands far off axis, where the
    # lens passes nothing and every visit burns its 5 x samples attempts.)
    frame_world = emulate[0] if emulate else world
    if frame_world == 1:
        W, Hr = args.width, args.height
    else:
        W = int(round(args.width * math.sqrt(frame_world)))
        Hr = int(round(args.height / math.sqrt(frame_world)))
    H = Hr * frame_world
    p = camera.default_params()
    camera.setup_filter(p, W, H, filter_width=1.0, aa_samples=3)
    p, model = camera.setup_po(p, args.lens, focus_dist=150.0)
    p.samples_override = args.samples
    if args.bokeh_image:
        p.bokeh_enable_image = 1
    table, keep = lens_io.make_lens_table(model.spec)
    tan_half_fov = float(p.sensor_width) * 0.5 / float(p.focal_length)

    ctx = capi.Context(local_rank)
    ctx.set_params(p)
    ctx.set_lens(table)
    if args.bokeh_image:
        from pota_amd import bokeh
        tex = np.load(os.path.join(ROOT, "tests", "golden", "example_bokeh_kernel_u8.npy")).astype(np.float32) / np.float32(255)
        ctx.set_bokeh(bokeh.build_tables(tex))
    ctx.alloc_frame(1 + args.aovs)
    engine = distributed.HipEngine(ctx, rows=p.yres)
    # N > 1: every rank owns a band of consecutive rows (its visits and its tile of the output) and sends the
    # rows its draws touched outside the band to their owners; LENTIL_PARTITION=interleaved selects rows
    # r mod N with one sum all-reduce over the whole frame instead
    tiled = (world > 1 or force_dist) and os.environ.get("LENTIL_PARTITION", "bands") != "interleaved"
    # band boundaries of the tiled mode: even split, --bounds, or (N > 1) re-cut after two calibration passes so that
    # every rank's pass takes the same time (distributed.rebalance; LENTIL_REBALANCE=0 keeps the even split)
    state = {"bounds": [int(x) for x in args.bounds.split(",")] if args.bounds else None, "band": None,
             "v_begin": 0, "v_end": 0}

    def set_band():
        if emulate:
            state["band"] = distributed.band_of(emulate[1], emulate[0], H, p.yres, state["bounds"])
        else:
            state["band"] = distributed.band_of(rank, world, H, p.yres, state["bounds"])
        state["v_begin"] = state["band"][0] * W * M
        state["v_end"] = min(state["band"][1], H) * W * M
        return state["v_end"] - state["v_begin"]

    if emulate:
        tiled = False
        n_local = set_band()
    elif tiled:
        n_local = set_band()
    else:
        n_local = workload.frame_visit_count(W, H, M, world, rank)
    bytes_per_visit = 80 + 16 * args.aovs

    def bind(f_hi):
        if tiled or emulate:
            cols = workload.generate(torch, state["v_begin"], state["v_end"], W, H, M, f_hi=f_hi, focus_dist=150.0,
                                     tan_half_fov=tan_half_fov, n_extra=args.aovs, device=dev)
            torch.cuda.synchronize()
            v, kv = capi.make_visits(cols, visits_per_pixel=M, pixels_per_row=W, pixel_y0=state["band"][0],
                                     ptr=lambda t: t.data_ptr())
        else:
            cols = workload.generate(torch, 0, n_local, W, H, M, f_hi=f_hi, focus_dist=150.0, tan_half_fov=tan_half_fov,
                                     n_extra=args.aovs, device=dev, row_stride=world, row_offset=rank)
            torch.cuda.synchronize()
            v, kv = capi.make_visits(cols, visits_per_pixel=M, pixels_per_row=W, pixel_y0=rank, pixel_row_stride=world,
                                     ptr=lambda t: t.data_ptr())
        ctx.bind_visits(v, kv)
        return cols

    def step():
        if tiled:
            distributed.frame_step_bands(engine, dist, H, p.yres, state["bounds"])
        else:
            distributed.frame_step(engine, dist)

    def run(steps, warmup):
        for _ in range(warmup):
            step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        scan_ms = draw_ms = res_ms = 0.0
        for _ in range(steps):
            step()
            # HIP-event times of this step's kernels (the call waits for the step's stream work, which
            # the step would have to finish anyway before the next clear)
            a, b, c = ctx.last_timing()
            scan_ms += a; draw_ms += b; res_ms += c
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, scan_ms / steps, draw_ms / steps, res_ms / steps

    cols = bind(args.f_hi)
    if tiled and (world > 1 or force_dist) and not args.bounds and os.environ.get("LENTIL_REBALANCE", "1") != "0":
        # calibration (untimed set-up, before the warm-up steps): three passes, the first with even bands, each followed
        # by an all-gather of the ranks' pass times (scan + draws, HIP events) and a re-cut of the bands
        state["bounds"] = distributed.even_bounds(world, H)
        for it in range(3):
            step()
            a, b, c = ctx.last_timing()
            mine = torch.tensor([a + b], dtype=torch.float64, device=dev)
            allt = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allt, mine)
            times = torch.stack(allt).flatten().tolist()
            new = distributed.rebalance(state["bounds"], times, damping=(1.0, 0.8, 0.6)[it])
            if new != state["bounds"]:
                state["bounds"] = new
                n_local = set_band()
                del cols
                torch.cuda.empty_cache()
                cols = bind(args.f_hi)
    else:
        # set-up: the first pass of a context sizes its draw buffers from the scan's counters (host round trips,
        # allocations); two passes settle that, whatever --warmup says
        for _ in range(2):
            step()
        torch.cuda.synchronize()
    dt, scan_ms, draw_ms, res_ms = run(args.steps, args.warmup)
    ctr = ctx.counters()
    n_total = workload.frame_visit_count(W, H, M) if not emulate else n_local        # all ranks
    value = n_total * args.steps / dt / 1e6
    ms_per_step = dt / args.steps * 1e3

    workload_tag = "%s %dx%d M=%d samples=%d aovs=%d f_hi=%.3g" % (args.lens, W, Hr, M, args.samples, 1 + args.aovs, args.f_hi)
    # the scan runs as one launch per chunk of the visit stream; bytes and duration below are per launch
    scan_launches = max(1, ctx.last_launches()[0])
    launch_bytes = n_local * bytes_per_visit / scan_launches
    launch_ms = scan_ms / scan_launches
    achieved = launch_bytes / (launch_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    out = {
        "metric": "bidir redistribution Msamples/s at 4K, double-gauss 50mm",
        "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": "polynomial-optics %s (self-fitted table), frame %dx%d (%d x the pixels of %dx%d, same camera), "
                        "%d rows per GPU, %d visits/pixel, %d redistribution draws per redistributed visit, %d AOV(s), "
                        "highlight fraction f_hi=%.3g"
                        % (args.lens, W, H, world, args.width, args.height, Hr, M, args.samples, 1 + args.aovs, args.f_hi),
            "visits_per_gpu": n_local, "bytes_per_visit": bytes_per_visit,
            "redistributed_visits_rank0": int(ctr.redistributed_visits),
            "attempted_draws_rank0": int(ctr.attempted_draws), "accepted_draws_rank0": int(ctr.accepted_draws),
            "parallelism": ("single GPU" if world == 1 else
                            "%d row bands%s, rows touched outside a band sent to its owner (p2p), tiled output"
                            % (world, (" at rows %s (balanced by pass time)" % state["bounds"]) if state["bounds"] else "") if tiled else
                            "rows%%%d + allreduce" % world),
        },
        "kernels_ms": {"scan": round(scan_ms, 4), "draw": round(draw_ms, 4), "resolve": round(res_ms, 4)},
        "roofline": {
            "kernel": "scan_uniform_kernel", "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": load_traffic(workload_tag),
            "launches_per_step": scan_launches, "algorithmic_bytes_per_launch": round(launch_bytes),
            "avg_launch_ms": round(launch_ms, 4),
            "whole_step_frac": round(n_local * bytes_per_visit / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
        },
        "draw_kernel": {"Mdraws_per_s_attempted": round(ctr.attempted_draws / (draw_ms * 1e-3) / 1e6, 3) if draw_ms > 0 else None},
    }

    if not args.no_second_regime and world == 1:
        del cols
        torch.cuda.empty_cache()
        f2 = 1.6e-3
        cols = bind(f2)
        st = max(1, min(args.steps, 2))
        dt2, s2, d2, r2 = run(st, 1)
        c2 = ctx.counters()
        out["regimes"] = {"highlight_heavy": {
            "f_hi": f2, "value": round(n_total * st / dt2 / 1e6, 3), "unit": "Msamples/s", "steps": st,
            "kernels_ms": {"scan": round(s2, 4), "draw": round(d2, 4), "resolve": round(r2, 4)},
            "attempted_draws": int(c2.attempted_draws),
            "Mdraws_per_s_attempted": round(c2.attempted_draws / (d2 * 1e-3) / 1e6, 3) if d2 > 0 else None}}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(args, p, table, M, tan_half_fov)
        except Exception as e:      # the GPU number must still be reported
            out["cpu_baseline"] = {"value": None, "unit": "Msamples/s", "cores": 0, "kind": "port",
                                   "sample": "failed: %r" % (e,)}
    if distributed.PHASE_SECONDS:
        sys.stderr.write("[band timing, ms per step incl. warm-up steps] %s\n" % {k: round(v * 1e3 / (args.steps + args.warmup), 3)
                                                                 for k, v in distributed.PHASE_SECONDS.items()})
    ctx.close()
    used_rccl = dist.is_initialized()
    if used_rccl:
        dist.destroy_process_group()
    if rank == 0:
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
