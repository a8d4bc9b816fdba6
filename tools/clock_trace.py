#!/usr/bin/env python3
"""The shader clock while the bench's steps run: a one-wave kernel (tools/micro/clock_sampler.hip, built on first use) samples
clock64() against the 100 MHz counter every 10 us beside the passes; every step's start is stamped in the same time base from the
context's main stream.  Prints, per step, the clock in 100-us bins from the step's start, and the means over the steps.
    python3 tools/clock_trace.py [--steps 12] [--lens ...] [--aovs N] [--width W --height H --samples S] [--f-hi F]
What it is for: the solves of a streamed pass run at 10-14 G lane-iterations/s beside the scan and for a while behind it, where the
same kernel at the same two waves per SIMD reaches 15.7 with the chip to itself (profiles/r06_fused_scan_ab.txt) -- is it the clock?"""
import argparse
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


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
    so = os.path.join(ROOT, "tools", "micro", "libclock_sampler.so")
    src = os.path.join(ROOT, "tools", "micro", "clock_sampler.hip")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", src, "-o", so])
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
    lib = C.CDLL(so)
    lib.sampler_start.argtypes = [C.c_uint32, C.c_uint32]
    lib.sampler_stamp.argtypes = [C.c_void_p, C.c_uint32]
    lib.sampler_read.argtypes = [C.c_void_p, C.c_void_p]
    n = int((a.steps + 4) * (ms_step + (0.3 if a.sync_each else 0.0)) * 1e3 / a.period_us) + 400
    assert lib.sampler_start(n, a.period_us * 100) == 0
    stream = b.ctx.stream()
    import time
    time.sleep(0.002)
    for k in range(a.steps):
        assert lib.sampler_stamp(C.c_void_p(stream), k) == 0
        b.step()
        if a.sync_each:
            b.ctx.sync()
    assert lib.sampler_stamp(C.c_void_p(stream), a.steps) == 0
    b.ctx.sync()
    pairs = (C.c_ulonglong * (2 * n))()
    stamps = (C.c_ulonglong * 4096)()
    assert lib.sampler_read(pairs, stamps) == 0
    t = [pairs[2 * i] for i in range(n)]
    mhz = [pairs[2 * i + 1] for i in range(n)]
    print("# %dx%d %s aovs %d samples %d f_hi %.3g: %.4f ms per step (timed run before the trace); one sample per %d us; %s"
          % (a.width, a.height, a.lens, a.aovs, a.samples, a.f_hi, ms_step, a.period_us,
             "every step waited for" if a.sync_each else "steps pipelined"))
    bin_us = 100
    n_bins = int(ms_step * 1e3 / bin_us) + 3
    acc = [[0, 0] for _ in range(n_bins)]
    for k in range(a.steps):
        t0, t1 = stamps[k], stamps[k + 1]
        row = [[0, 0] for _ in range(n_bins)]
        for ti, m in zip(t, mhz):
            if t0 <= ti < t1:
                bidx = int((ti - t0) / 100 / bin_us)
                if bidx < n_bins:
                    row[bidx][0] += m; row[bidx][1] += 1
        print("step %2d (%.3f ms):" % (k, (t1 - t0) / 1e5), " ".join("%4d" % (s // c) if c else "   -" for s, c in row))
        if k >= 2:
            for i, (s, c) in enumerate(row):
                acc[i][0] += s; acc[i][1] += c
    print("mean from step 2 on, MHz per %d-us bin from the step's start:" % bin_us)
    print("                   ", " ".join("%4d" % (s // c) if c else "   -" for s, c in acc))
    tot = sum(s for s, c in acc); cnt = sum(c for s, c in acc)
    print("mean over the steps: %d MHz; idle before the first step: %d MHz" % (tot // max(cnt, 1), sum(mhz[:50]) // 50))
    b.streams = []
    b.ctx.close()


if __name__ == "__main__":
    main()
