#!/usr/bin/env python3
"""How the step time of the bench's workload develops over seconds of back-to-back passes: chunks of `--chunk` steps timed one after the
other in one process (Bench.run: nothing waits inside a chunk), the clock in the pass sampled at the start and at the end
(tools/clock_trace.py).  A 20-step bench run lives in the first 40 ms; the chip's power management moves between regimes over tenths of a second to
seconds (profiles/r06_sustained.txt: 2.0 ms a step at ~2.0 GHz in the solve phase, 1.86 ms at ~2.25 GHz).
    python3 tools/sustained.py [--chunks 30] [--chunk 100] [--lens ...] [--aovs N] [--width W --height H --samples S]"""
import argparse
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=30)
    ap.add_argument("--chunk", type=int, default=100)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--samples", type=int, default=1024)
    ap.add_argument("--lens", default="double_gauss_50mm")
    ap.add_argument("--aovs", type=int, default=0)
    ap.add_argument("--f-hi", type=float, default=2.0 ** -16)
    a = ap.parse_args()
    import time
    import torch
    import bench
    import clock_trace
    dev = torch.device("cuda:0")
    b = bench.Bench(torch, None, dev, 0, 1, 0, a.width, a.height, a.height, 9, a.lens, a.samples, a.aovs, a.f_hi, False)
    b.generate(a.f_hi)
    for _ in range(4):
        b.step()
    b.ctx.sync()
    time.sleep(0.5)                      # (an idle chip, as a fresh bench run finds it)
    r0 = b.run(20, 3)
    ms20 = r0["dt"] / r0["steps"] * 1e3
    print("# %dx%d %s aovs %d samples %d: 20 steps after 3 warm-up steps on an idle chip: %.4f ms per step" % (a.width, a.height, a.lens, a.aovs, a.samples, ms20))
    c0 = clock_trace.clock_in_pass(b, ms20)
    print("# clock in the pass right after: mean %.0f MHz, middle half of a step %.0f MHz" % (c0["mean_mhz"], c0["middle_half_of_a_step_mhz"]))
    t_total = 0.0
    for k in range(a.chunks):
        r = b.run(a.chunk, 0)
        ms = r["dt"] / r["steps"] * 1e3
        t_total += r["dt"]
        print("chunk %2d (steps %5d-%5d, %.2f s in): %.4f ms per step   scan %.4f draw %.4f" % (k, k * a.chunk, (k + 1) * a.chunk - 1, t_total, ms, r["scan"] / r["steps"], r["draw"] / r["steps"]))
    c1 = clock_trace.clock_in_pass(b, ms)
    print("# clock in the pass at the end: mean %.0f MHz, middle half of a step %.0f MHz" % (c1["mean_mhz"], c1["middle_half_of_a_step_mhz"]))
    b.streams = []
    b.ctx.close()


if __name__ == "__main__":
    main()
