"""Development aid: the FIRST streamed pass of a context (the one that also calibrates the first-batch model), many contexts in a
row on the headline frame, in the order of tests/test_gpu_headline.py::test_headline_4k_streamed_vs_oracle (two seeded streams,
a draw log, the frame and the log downloaded between the passes) -- does it ever fall back or stall, and why
(lentil_hip_last_redo_note)?  usage: python3 tools/first_streamed_soak.py [contexts] [seconds of host work between passes]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from pota_amd import camera, capi, lens_io, workload
n_ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 30
idle_s = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
W, H, M = 3840, 2160, 9
dev = torch.device("cuda:0")
p = camera.default_params(); camera.setup_filter(p, W, H, filter_width=1.0, aa_samples=3)
p, model = camera.setup_po(p, "double_gauss_50mm", focus_dist=150.0); p.samples_override = 1024
table, keep = lens_io.make_lens_table(model.spec)
n = W * H * M
streams = []
for seed in (0x5EED, 0xBEEF):
    cols = workload.generate(torch, 0, n, W, H, M, seed=seed, f_hi=2.0 ** -16, focus_dist=150.0,
                             tan_half_fov=float(p.sensor_width) * 0.5 / float(p.focal_length), device=dev)
    streams.append((cols, capi.make_visits(cols, visits_per_pixel=M, pixels_per_row=W, ptr=lambda t: t.data_ptr())))
torch.cuda.synchronize()
bad = 0
for k in range(n_ctx):
    ctx = capi.Context(0); ctx.set_params(p); ctx.set_lens(table); ctx.set_bokeh(None); ctx.alloc_frame(1); ctx.set_draw_log(1 << 21)
    out = []
    for i in (0, 1, 0, 1):
        visits, kv = streams[i][1]
        ctx.bind_visits(visits, kv)
        t0 = time.perf_counter(); ctx.clear_frame(); ctx.redistribute(); ctx.resolve(); ctx.sync(); dt = (time.perf_counter() - t0) * 1e3
        c = ctx.counters()
        out.append((int(c.streamed), int(c.fallback_chunks), round(dt, 2)))
        ctx.draw_log(); ctx.download_accum(0); ctx.download_aov(0)
        if idle_s > 0.0:
            time.sleep(idle_s)
    ok = all(o[:2] == (1, 0) for o in out[1:])
    bad += 0 if ok else 1
    print(k, out, "" if ok else "  <-- NOT STREAMED / REDONE: " + ctx.last_redo_note(), flush=True)
    ctx.close()
print("contexts with a pass that was not streamed or was redone:", bad, "of", n_ctx)
