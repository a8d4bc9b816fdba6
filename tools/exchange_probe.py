#!/usr/bin/env python3
"""Host-side cost of lentil_hip_exchange_bands in its two forms, on ONE GPU: `world` ranks as threads of this process, each
with a context and a band of the 4K headline frame, tests/fake_rccl/libfake_rccl.so standing in for RCCL (device-to-device
copies inside the process).  What this can show is the exchange's own structure -- host waits, launches, bytes -- not xGMI:
the ranks share one GPU, so pass times are not meaningful and only the exchange (barrier -> exchange_bands -> sync) is timed.

    python3 tools/exchange_probe.py [world] [steps]        # runs itself twice: LENTIL_EXCHANGE_FIXED=0 and =1
"""
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def child(world, steps):
    import numpy as np
    import torch
    import common
    from pota_amd import capi, distributed, workload
    W, H, M, S, f_hi = 3840, 2160, 9, 1024, 2.0 ** -16
    p, model, table, keep = common.po_setup(W, H, samples_override=S)
    dev = torch.device("cuda:0")
    ctxs, keepalive, bands = [], [], []
    for rank in range(world):
        b_lo, b_hi = distributed.band_of(rank, world, H, p.yres, None)
        cols = workload.generate(torch, b_lo * W * M, min(b_hi, H) * W * M, W, H, M, f_hi=f_hi, focus_dist=150.0,
                                 tan_half_fov=common.tan_half_fov(p), device=dev)
        v, kv = capi.make_visits(cols, visits_per_pixel=M, pixels_per_row=W, pixel_y0=b_lo, ptr=lambda t: t.data_ptr())
        ctx = capi.Context(0)
        ctx.set_params(p); ctx.set_lens(table); ctx.set_bokeh(None); ctx.alloc_frame(1)
        ctx.bind_visits(v, kv)
        keepalive.append((cols, v, kv)); ctxs.append(ctx); bands.append((b_lo, b_hi))
    torch.cuda.synchronize()
    uid = capi.Context.comm_unique_id()
    bar = threading.Barrier(world)
    times = [[] for _ in range(world)]
    stats = [None] * world

    def rank_fn(rank):
        ctx = ctxs[rank]
        ctx.comm_init(uid, rank, world)
        for k in range(steps + 2):
            bar.wait()
            ctx.set_closest_exchange(False)
            ctx.clear_frame(); ctx.redistribute(); ctx.sync()
            bar.wait()
            t0 = time.perf_counter()
            ctx.exchange_bands(H, None, sparse=True); ctx.sync()
            t1 = time.perf_counter()
            if k >= 2:
                times[rank].append((t1 - t0) * 1e3)
        stats[rank] = (ctx.exchange_stats(), ctx.exchange_counts())
        ctx.comm_destroy()

    th = [threading.Thread(target=rank_fn, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    print(json.dumps({"form": "fixed" if os.environ.get("LENTIL_EXCHANGE_FIXED", "1") != "0" else "sized", "world": world,
                      "exchange_ms_per_rank_median": [round(sorted(t)[len(t) // 2], 3) for t in times],
                      "sent_received_bytes": [s[0] for s in stats], "fixed_overflow": [s[1] for s in stats]}))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]))
    else:
        world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
        steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
        for fixed in ("0", "1"):
            env = dict(os.environ, LENTIL_EXCHANGE_FIXED=fixed, LENTIL_RCCL_LIB=os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so"))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(world), str(steps)], env=env,
                               capture_output=True, text=True)
            sys.stdout.write(r.stdout[-2000:])
            if r.returncode:
                sys.stderr.write(r.stderr[-3000:])
