#!/usr/bin/env python3
"""The shader clock while the bench's steps run: a one-wave kernel (tools/micro/clock_sampler.hip, built by __graft_entry__.build()
or on first use) samples clock64() against the 100 MHz counter every 10 us beside the passes; every step's start is stamped in the
same time base from the context's main stream.  Prints, per step, the clock in 100-us bins from the step's start, and the means
over the steps.
    python3 tools/clock_trace.py [--steps 12] [--lens ...] [--aovs N] [--width W --height H --samples S] [--f-hi F] [--sync-each]
What it is for: the solves of a streamed pass run at 10-14 G lane-iterations/s beside the scan and for a while behind it, where the
same kernel at the same two waves per SIMD reaches 15.7 with the chip to itself (profiles/r06_fused_scan_ab.txt) -- is it the clock?
(It is: DESIGN.md section 4.2c.)  bench.py calls clock_in_pass() for its line's roofline.clock_in_pass."""
import argparse
import ctypes as C
import os
import subprocess
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def load_sampler():
    so = os.path.join(ROOT, "tools", "micro", "libclock_sampler.so")
    src = os.path.join(ROOT, "tools", "micro", "clock_sampler.hip")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", src, "-o", so])
    lib = C.CDLL(so)
    lib.sampler_start.argtypes = [C.c_uint32, C.c_uint32]
    lib.sampler_stamp.argtypes = [C.c_void_p, C.c_uint32]
    lib.sampler_read.argtypes = [C.c_void_p, C.c_void_p]
    return lib


def trace_steps(b, steps, ms_step, period_us=10, sync_each=False, bin_us=100):
    """Run `steps` steps of the Bench object `b` with the sampler beside them.  Returns (rows, idle_mhz): per step (its length in
    ms, the list of [sum, count] of the sampled MHz per `bin_us` bin from the step's start), and the clock before the first step."""
    lib = load_sampler()
    n = int((steps + 4) * (ms_step + (0.3 if sync_each else 0.0)) * 1e3 / period_us) + 400
    if lib.sampler_start(n, period_us * 100) != 0:
        raise RuntimeError("clock sampler: start failed")
    stream = b.ctx.stream()
    time.sleep(0.002)
    for k in range(steps):
        if lib.sampler_stamp(C.c_void_p(stream), k) != 0:
            raise RuntimeError("clock sampler: stamp failed")
        b.step()
        if sync_each:
            b.ctx.sync()
    lib.sampler_stamp(C.c_void_p(stream), steps)
    b.ctx.sync()
    pairs = (C.c_ulonglong * (2 * n))()
    stamps = (C.c_ulonglong * 4096)()
    if lib.sampler_read(pairs, stamps) != 0:
        raise RuntimeError("clock sampler: read failed")
    t = [pairs[2 * i] for i in range(n)]
    mhz = [pairs[2 * i + 1] for i in range(n)]
    n_bins = int(ms_step * 1e3 / bin_us) + 3
    rows = []
    for k in range(steps):
        t0, t1 = stamps[k], stamps[k + 1]
        row = [[0, 0] for _ in range(n_bins)]
        for ti, m in zip(t, mhz):
            if t0 <= ti < t1:
                bidx = int((ti - t0) / 100 / bin_us)
                if bidx < n_bins:
                    row[bidx][0] += m
                    row[bidx][1] += 1
        rows.append(((t1 - t0) / 1e5, row))
    return rows, sum(mhz[:50]) // 50


def clock_in_pass(b, ms_step, steps=10):
    """What bench.py reports as roofline.clock_in_pass: the shader clock while the timed workload's steps run (a short loop of its
    own with the sampler beside it, after the timed one), the mean over the steps and over the middle half of a step."""
    rows, idle = trace_steps(b, steps, ms_step)
    tot = cnt = mid_tot = mid_cnt = 0
    for k, (ms, row) in enumerate(rows):
        if k < 2:
            continue
        filled = [i for i, (s, c) in enumerate(row) if c]
        if not filled:
            continue
        lo, hi = filled[0], filled[-1] + 1
        q = (hi - lo) // 4
        for i, (s, c) in enumerate(row):
            tot += s
            cnt += c
            if lo + q <= i < hi - q:
                mid_tot += s
                mid_cnt += c
    return {"mean_mhz": round(tot / max(cnt, 1), 1), "middle_half_of_a_step_mhz": round(mid_tot / max(mid_cnt, 1), 1),
            "idle_mhz": int(idle),
            "how": "tools/micro/clock_sampler.hip: one wave sampling clock64() against the 100 MHz counter every 10 us beside %d "
                   "steps of the timed workload (a loop of its own, after the timed one), every step's start stamped from the "
                   "context's stream; DESIGN.md section 4.2c says why this, not box.shader_clock_mhz_under_fp64 (the plateau "
                   "after 15 ms of steady fp64 load), is the clock a 2-ms pass runs at" % steps}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--samples", type=int, default=1024)
    ap.add_argument("--lens", default="double_gauss_50mm")
    ap.add_argument("--aovs", type=int, default=0)
    ap.add_argument("--f-hi", type=float, default=2.0 ** -16)
    ap.add_argument("--period-us", type=int, default=10)
    ap.add_argument("--sync-each", action="store_true", help="wait for every step's end before the next (a pause between the passes)")
    a = ap.parse_args()
    import torch
    import bench
    dev = torch.device("cuda:0")
    b = bench.Bench(torch, None, dev, 0, 1, 0, a.width, a.height, a.height, 9, a.lens, a.samples, a.aovs, a.f_hi, False)
    b.generate(a.f_hi)
    for _ in range(4):
        b.step()
    b.ctx.sync()
    r = b.run(8, 2)
    ms_step = r["dt"] / r["steps"] * 1e3
    rows, idle = trace_steps(b, a.steps, ms_step, a.period_us, a.sync_each)
    print("# %dx%d %s aovs %d samples %d f_hi %.3g: %.4f ms per step (timed run before the trace); one sample per %d us; %s"
          % (a.width, a.height, a.lens, a.aovs, a.samples, a.f_hi, ms_step, a.period_us,
             "every step waited for" if a.sync_each else "steps pipelined"))
    n_bins = max(len(row) for _, row in rows)
    acc = [[0, 0] for _ in range(n_bins)]
    for k, (ms, row) in enumerate(rows):
        print("step %2d (%.3f ms):" % (k, ms), " ".join("%4d" % (s // c) if c else "   -" for s, c in row))
        if k >= 2:
            for i, (s, c) in enumerate(row):
                acc[i][0] += s
                acc[i][1] += c
    print("mean from step 2 on, MHz per 100-us bin from the step's start:")
    print("                   ", " ".join("%4d" % (s // c) if c else "   -" for s, c in acc))
    tot = sum(s for s, c in acc)
    cnt = sum(c for s, c in acc)
    print("mean over the steps: %d MHz; idle before the first step: %d MHz" % (tot // max(cnt, 1), idle))
    b.streams = []
    b.ctx.close()


if __name__ == "__main__":
    main()
