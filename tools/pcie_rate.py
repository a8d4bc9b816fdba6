"""What a frame costs when the visit columns start in host memory: (a) one upload at the frame end
(lentil_hip_upload_visits, pageable memory), (b) the same from page-locked memory, (c) handed over in blocks from
page-locked memory while "the render" produces them (lentil_hip_visits_begin / _append / _end) -- there the figure
that matters is what the frame end still waits for after the last block.  Writes one JSON object (also to
gpurun_out/pcie_rate.json).  usage: pcie_rate.py [W H] [render_seconds]"""
import ctypes as C
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch  # noqa: F401  (one HIP runtime per process)
from pota_amd import _abi, camera, capi, lens_io, workload

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
RENDER_S = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
M = 9
p = camera.default_params(); camera.setup_filter(p, W, H, filter_width=1.0, aa_samples=3)
p, model = camera.setup_po(p, "double_gauss_50mm", focus_dist=150.0); p.samples_override = 1024
table, keep = lens_io.make_lens_table(model.spec)
n = W * H * M
cols = workload.generate(np, 0, n, W, H, M, f_hi=2.0 ** -16, focus_dist=150.0, tan_half_fov=float(p.sensor_width) * 0.5 / float(p.focal_length))
visits, kv = capi.make_visits(cols, visits_per_pixel=M, pixels_per_row=W)
ctx = capi.Context(0); ctx.set_params(p); ctx.set_lens(table); ctx.alloc_frame(1)
nbytes = n * 80
NAMES = ("rgba", "pos_z", "raydir_time", "volume_ignore", "transmission")
out = {"frame": "%dx%d M=%d" % (W, H, M), "visits": n, "bytes": nbytes}


def run_pass():
    ctx.clear_frame(); ctx.redistribute(); ctx.resolve(); ctx.sync()


def best(fn, reps=3):
    r = [fn() for _ in range(reps)]
    return min(r, key=lambda x: x[0])


def whole(v):
    t0 = time.perf_counter(); ctx.upload_visits(v); ctx.sync(); t1 = time.perf_counter()
    run_pass(); t2 = time.perf_counter()
    return (t2 - t0, t1 - t0, t2 - t1)


tot, up, ps = best(lambda: whole(visits))
out["one_upload_pageable"] = {"upload_ms": round(up * 1e3, 2), "GBps": round(nbytes / up / 1e9, 1), "pass_ms": round(ps * 1e3, 3),
                              "Mvisits_per_s_upload_plus_pass": round(n / tot / 1e6, 1)}
out["pass_alone_Mvisits_per_s"] = round(n / ps / 1e6, 1)

# the same columns in page-locked memory
pin_ptr = capi.host_alloc(nbytes)
pin = np.frombuffer((C.c_char * nbytes).from_address(pin_ptr), np.float32).reshape(5, n, 4)
for c, name in enumerate(NAMES):
    pin[c] = cols[name]
pcols = {name: pin[c] for c, name in enumerate(NAMES)}
pcols["extra"] = []
pvisits, pkv = capi.make_visits(pcols, visits_per_pixel=M, pixels_per_row=W)
tot, up, ps = best(lambda: whole(pvisits))
out["one_upload_pinned"] = {"upload_ms": round(up * 1e3, 2), "GBps": round(nbytes / up / 1e9, 1), "pass_ms": round(ps * 1e3, 3),
                            "Mvisits_per_s_upload_plus_pass": round(n / tot / 1e6, 1)}

# handed over in blocks while the render runs: `threads` producers, each owning a share of the rows, pace themselves so
# that the last block leaves at RENDER_S; blocks are slices of the page-locked copy (a capturing filter_pixel writes
# them there in the first place)
BLOCK = 1 << 16
layout = _abi.Visits()
layout.visits_per_pixel, layout.pixels_per_row, layout.pixel_row_stride = M, W, 1


def streamed(render_s):
    ctx.visits_begin(layout, n)
    t0 = time.perf_counter()
    cuts = list(range(0, n, BLOCK)) + [n]
    for j in range(len(cuts) - 1):           # in order (a uniform stream is pixel-major): one producer
        if render_s:
            due = t0 + render_s * (j + 1) / (len(cuts) - 1)
            while time.perf_counter() < due:
                time.sleep(0.0002)
        lo, hi = cuts[j], cuts[j + 1]
        part = _abi.Visits()
        part.n = hi - lo
        for c, name in enumerate(NAMES):
            setattr(part, name, pin[c, lo:hi].ctypes.data)
        ctx.visits_append(part)
    t1 = time.perf_counter()                 # "the render" is over: what the frame end still waits for
    ctx.visits_end()
    t2 = time.perf_counter()
    run_pass()
    t3 = time.perf_counter()
    return (t3 - t1, t1 - t0, t2 - t1, t3 - t2)


tail, feed, end, ps = best(lambda: streamed(0.0))
out["blocks_back_to_back"] = {"block_visits": BLOCK, "append_calls_ms": round(feed * 1e3, 2), "visits_end_ms": round(end * 1e3, 2),
                              "GBps": round(nbytes / (feed + end) / 1e9, 1), "pass_ms": round(ps * 1e3, 3),
                              "Mvisits_per_s_upload_plus_pass": round(n / (feed + end + ps) / 1e6, 1)}
tail, feed, end, ps = best(lambda: streamed(RENDER_S), reps=2)
out["blocks_during_a_render"] = {"render_s": RENDER_S, "block_visits": BLOCK, "visits_end_after_last_block_ms": round(end * 1e3, 3),
                                 "pass_ms": round(ps * 1e3, 3), "frame_end_wait_ms": round(tail * 1e3, 3),
                                 "Mvisits_per_s_of_the_frame_end": round(n / tail / 1e6, 1)}
capi.host_free(pin_ptr)
s = json.dumps(out)
print(s)
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/pcie_rate.json", "w").write(s + "\n")
