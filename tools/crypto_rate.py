"""What cryptomatte AOVs add to the headline frame (3840x2160, 9 visits per pixel, double gauss): the pass without
them, then with 1 and 3 cryptomatte AOVs of `entries` ids per visit -- the replay kernels (own-pixel visits, draw log)
run behind the pass on its stream.  Columns are generated on the device.  Writes one JSON object (also to
gpurun_out/crypto_rate.json).  usage: crypto_rate.py [W H] [entries]"""
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # (a fifth hardware queue: the replay's first half runs beside the draws)

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from pota_amd import camera, capi, lens_io, workload

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
ENTRIES = int(sys.argv[3]) if len(sys.argv) > 3 else 2
M = 9
dev = torch.device("cuda:0")
p = camera.default_params(); camera.setup_filter(p, W, H, filter_width=1.0, aa_samples=3)
p, model = camera.setup_po(p, "double_gauss_50mm", focus_dist=150.0); p.samples_override = 1024
table, keep = lens_io.make_lens_table(model.spec)
n = W * H * M
cols = workload.generate(torch, 0, n, W, H, M, f_hi=2.0 ** -16, focus_dist=150.0, tan_half_fov=float(p.sensor_width) * 0.5 / float(p.focal_length), device=dev)
visits, kv = capi.make_visits(cols, visits_per_pixel=M, pixels_per_row=W, ptr=lambda t: t.data_ptr())
out = {"frame": "%dx%d M=%d" % (W, H, M), "visits": n, "entries": ENTRIES}


def timed(ctx, reps=5):
    best = 1e9
    for _ in range(reps):
        ctx.clear_frame(); ctx.sync()
        t0 = time.perf_counter(); ctx.redistribute(); ctx.sync(); t1 = time.perf_counter()
        best = min(best, t1 - t0)
    return best * 1e3


def timed_with_clear(ctx, reps=5):
    """lentil_hip_clear_frame inside the timed region: what a frame costs end to end (round 5: the wipe of the cryptomatte tables
    is put off to the pass, so the pass alone is no longer the whole story)"""
    best = 1e9
    for _ in range(reps):
        ctx.sync()
        t0 = time.perf_counter(); ctx.clear_frame(); ctx.redistribute(); ctx.sync(); t1 = time.perf_counter()
        best = min(best, t1 - t0)
    return best * 1e3


ONLY = os.environ.get("LENTIL_CRYPTO_RATE_ONLY")         # e.g. "0,1": just those variants (for a kernel trace of one of them)
for n_crypto in ([int(x) for x in ONLY.split(",")] if ONLY else (0, -1, 1, 3)):
    ctx = capi.Context(0); ctx.set_params(p); ctx.set_lens(table); ctx.alloc_frame(1)
    ctx.bind_visits(visits, kv)
    if n_crypto < 0:
        # the pass with a draw log and nothing else: what the log alone costs it
        ctx.set_draw_log(4 << 20)
        out["log_only"] = {"redistribute_ms": round(timed(ctx), 3), "streamed": int(ctx.counters().streamed), "timing": [round(x, 3) for x in ctx.last_timing()]}
        ctx.close()
        continue
    keepc = None
    if n_crypto:
        ctx.alloc_crypto(n_crypto, 16)
        pix_x = (torch.arange(n, device=dev) // M) % W
        hs, ws = [], []
        for a in range(n_crypto):
            g = torch.Generator(device=dev); g.manual_seed(a + 1)
            ids = torch.stack([((pix_x // 40 + a) * 7 + e).to(torch.float32) for e in range(ENTRIES)], 1).contiguous()
            w = torch.rand((n, ENTRIES), device=dev, generator=g)
            w = (w / w.sum(1, keepdim=True)).contiguous()
            hs.append(ids); ws.append(w)
        cv, keepc = capi.make_crypto_visits(hs, ws, ptr=lambda t: t.data_ptr())
        ctx.bind_crypto(cv, keepc)
        torch.cuda.synchronize()
    ms = timed(ctx)
    ms_c = timed_with_clear(ctx)
    c = ctx.counters()
    key = "crypto_%d" % n_crypto
    out[key] = {"redistribute_ms": round(ms, 3), "clear_and_redistribute_ms": round(ms_c, 3), "accepted_draws": int(c.accepted_draws),
                "streamed": int(c.streamed), "timing": [round(x, 3) for x in ctx.last_timing()]}
    if n_crypto:
        out[key]["added_ms"] = round(ms - out["crypto_0"]["redistribute_ms"], 3)
        out[key]["added_ms_per_aov"] = round((ms - out["crypto_0"]["redistribute_ms"]) / n_crypto, 3)
        out[key]["added_ms_per_aov_with_clear"] = round((ms_c - out["crypto_0"]["clear_and_redistribute_ms"]) / n_crypto, 3)
        # algorithmic bytes of the replay: which visits were redistributed (a bit each), per AOV the visits' pairs
        # (entries * 8 B) and the pixel's table lines (slots * 8 B + the total)
        alg = n // 8 + n_crypto * (n * ENTRIES * 8 + W * H * (16 * 8 + 4))      # (one bit per visit; tables written, not read, after a clear)
        out[key]["replay_GBps_algorithmic"] = round(alg / max(ms - out["crypto_0"]["redistribute_ms"], 1e-6) / 1e6, 1)
        t0 = time.perf_counter(); img, has = ctx.download_crypto(0, 0); t1 = time.perf_counter()
        out[key]["rank_and_download_ms"] = round((t1 - t0) * 1e3, 2)
    ctx.close()
    del keepc
s = json.dumps(out)
print(s)
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/crypto_rate.json", "w").write(s + "\n")
