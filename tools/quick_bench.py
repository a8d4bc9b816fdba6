"""Ad-hoc timing of the kernels (development aid; the contract benchmark is bench.py)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from pota_amd import camera, capi, lens_io, workload

W, H, M = int(sys.argv[1]), int(sys.argv[2]), 9
override = int(sys.argv[3]); f_hi = float(sys.argv[4]); K = int(sys.argv[5]) if len(sys.argv) > 5 else 0
p = camera.default_params(); camera.setup_filter(p, W, H, filter_width=1.0, aa_samples=3)
p, model = camera.setup_po(p, "double_gauss_50mm", focus_dist=150.0); p.samples_override = override
table, keep = lens_io.make_lens_table(model.spec)
dev = torch.device("cuda:0")
n = W * H * M
cols = workload.generate(torch, 0, n, W, H, M, f_hi=f_hi, focus_dist=150.0, tan_half_fov=float(p.sensor_width) * 0.5 / float(p.focal_length), n_extra=K, device=dev)
torch.cuda.synchronize()
ctx = capi.Context(0)
ctx.set_params(p); ctx.set_lens(table); ctx.alloc_frame(1 + K)
visits, kv = capi.make_visits(cols, visits_per_pixel=M, pixels_per_row=W, ptr=lambda t: t.data_ptr())
ctx.bind_visits(visits, kv)
for it in range(4):
    ctx.clear_frame(); ctx.redistribute(); ctx.resolve(); ctx.sync()
    ms = ctx.last_timing(); c = ctx.counters()
    bytes_ = n * (80 + 16 * K)
    print("iter %d: scan %.3f ms (%.1f GB/s) draw %.3f ms (%.2f Mdraws/s att) resolve %.3f ms | redis %d att %d acc %d"
          % (it, ms[0], bytes_ / ms[0] / 1e6, ms[1], c.attempted_draws / max(ms[1], 1e-9) / 1e3, ms[2],
             c.redistributed_visits, c.attempted_draws, c.accepted_draws))
    if c.tries:
        print("        tries/attempt %.2f iters/try %.1f lane-utilisation %.2f  -> %.1f G lane-iters/s"
              % (c.tries / c.attempted_draws, c.newton_iterations / c.tries, c.newton_iterations / max(c.lane_rounds, 1),
                 c.newton_iterations / ms[1] / 1e6))
