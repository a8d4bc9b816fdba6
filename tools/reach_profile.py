"""Development aid: how densely a band's draws fill the rows they touch outside the band (tiled multi-GPU mode).
usage: reach_profile.py N r   -- rank r's band of the N-GPU weak-scaling frame of bench.py, on one GPU."""
import math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from pota_amd import camera, capi, distributed, lens_io, workload

N, r = int(sys.argv[1]), int(sys.argv[2])
W = int(round(3840 * math.sqrt(N))); Hr = int(round(2160 / math.sqrt(N))); H = Hr * N; M = 9
p = camera.default_params(); camera.setup_filter(p, W, H, filter_width=1.0, aa_samples=3)
p, model = camera.setup_po(p, "double_gauss_50mm", focus_dist=150.0); p.samples_override = 1024
table, keep = lens_io.make_lens_table(model.spec)
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ctx = capi.Context(0); ctx.set_params(p); ctx.set_lens(table); ctx.alloc_frame(1)
lo, hi = distributed.band_of(r, N, H, p.yres)
cols = workload.generate(torch, lo * W * M, min(hi, H) * W * M, W, H, M, f_hi=2.0 ** -16, focus_dist=150.0,
                         tan_half_fov=float(p.sensor_width) * 0.5 / float(p.focal_length), device=dev)
torch.cuda.synchronize()
v, kv = capi.make_visits(cols, visits_per_pixel=M, pixels_per_row=W, pixel_y0=lo, ptr=lambda t: t.data_ptr())
ctx.bind_visits(v, kv); ctx.clear_frame(); ctx.redistribute(); ctx.sync()
eng = distributed.HipEngine(ctx, rows=p.yres)
t_lo, t_hi = ctx.touched_rows()
rec = eng.accum.view(p.yres, p.xres, -1)
nz = (rec[:, :, 4] != 0).sum(dim=1).cpu().numpy()      # pixels with weight per row
print("band", lo, hi, "touched", t_lo, t_hi, "xres", p.xres)
for name, rows in (("above", range(lo - 1, t_lo - 1, -1)), ("below", range(hi, t_hi))):
    rows = list(rows)
    if not rows: continue
    d = nz[rows]
    tot = d.sum()
    print(name, "rows", len(rows), "nonzero pixels", int(tot), "= %.1f %% of the rows' pixels" % (100.0 * tot / (len(rows) * p.xres)))
    cum = np.cumsum(d) / max(tot, 1)
    for q in (0.5, 0.9, 0.99):
        print("   %.0f %% of them within %d rows of the band" % (q * 100, int(np.searchsorted(cum, q)) + 1))
