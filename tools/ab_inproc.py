#!/usr/bin/env python3
"""A/B of environment settings on ONE box in ONE process: the headline frame (or --width/--height/--samples/--lens/--aovs), one
context per setting and repetition (created under that setting's environment -- knobs the library reads in lentil_hip_create or per
pass; a knob it caches in a function static cannot be varied this way -- warmed up, timed, destroyed), the two resident visit
streams shared, the settings' timed runs interleaved rep by rep (boxes drift by a few per cent over a minute; runs a second
apart do not).
    python3 tools/ab_inproc.py [--reps 6] [--steps 30] "LENTIL_X=1 LENTIL_Y=0" "LENTIL_X=0" ...      ("" = the defaults)
Prints every run and, per setting, mean / median / min of ms per step and of the event-timed kernels."""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("settings", nargs="+")
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--samples", type=int, default=1024)
    ap.add_argument("--lens", default="double_gauss_50mm")
    ap.add_argument("--aovs", type=int, default=0)
    ap.add_argument("--f-hi", type=float, default=2.0 ** -16)
    a = ap.parse_args()
    import torch
    import bench
    dev = torch.device("cuda:0")

    def env_of(text):
        return dict(kv.split("=", 1) for kv in text.split()) if text.strip() else {}

    class Env:
        def __init__(self, e):
            self.e = e
        def __enter__(self):
            self.old = {k: os.environ.get(k) for k in self.e}
            os.environ.update(self.e)
        def __exit__(self, *x):
            for k, v in self.old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v

    # One context at a time: a context's streams are spread over the runtime's hardware queues as they are created, and the
    # streams of several live contexts end up sharing queues -- the later contexts' passes then run with less overlap (seen: the
    # same setting 2.0 ms as the first context of a process and 2.4 ms as its fourth).  So: create, warm up, time, destroy.
    streams = None
    rows = {t: [] for t in a.settings}
    for rep in range(a.reps):
        for text in a.settings:
            with Env(env_of(text)):
                b = bench.Bench(torch, None, dev, 0, 1, 0, a.width, a.height, a.height, 9, a.lens, a.samples, a.aovs, a.f_hi, False)
                if streams is None:
                    b.generate(a.f_hi)
                    streams = b.streams
                else:
                    b.streams, b.i_stream = streams, 0
                for _ in range(3):                      # the context's first passes size its buffers
                    b.step()
                b.ctx.sync()
                r = b.run(a.steps, a.warmup)
                b.streams = []
                b.ctx.close()
            ms = r["dt"] / r["steps"] * 1e3
            rows[text].append((ms, r["scan"] / r["steps"], r["draw"] / r["steps"], r["resolve"] / r["steps"], r.get("rounds_max"), r.get("redone"), r.get("slow", 0) // r["steps"]))
            print("rep %d  %-60s %.4f ms   scan %.4f draw %.4f resolve %.4f  rounds %s redone %s parked %s" % ((rep, text or "(defaults)") + rows[text][-1]), flush=True)
    print()
    for text in a.settings:
        ms = [x[0] for x in rows[text]]
        kern = [x[1] + x[2] for x in rows[text]]
        print("%-60s ms/step mean %.4f median %.4f min %.4f max %.4f | scan+draw mean %.4f | outside the kernels %.4f"
              % (text or "(defaults)", statistics.mean(ms), statistics.median(ms), min(ms), max(ms), statistics.mean(kern),
                 statistics.mean(ms) - statistics.mean(kern)))


if __name__ == "__main__":
    main()
