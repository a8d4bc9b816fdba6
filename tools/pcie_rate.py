"""Development aid: what a frame costs when the visit columns start in host memory (lentil_hip_upload_visits):
upload + pass, against the pass alone.  usage: pcie_rate.py [W H]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch  # noqa: F401  (one HIP runtime per process)
from pota_amd import camera, capi, lens_io, workload

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
M = 9
p = camera.default_params(); camera.setup_filter(p, W, H, filter_width=1.0, aa_samples=3)
p, model = camera.setup_po(p, "double_gauss_50mm", focus_dist=150.0); p.samples_override = 1024
table, keep = lens_io.make_lens_table(model.spec)
n = W * H * M
cols = workload.generate(np, 0, n, W, H, M, f_hi=2.0 ** -16, focus_dist=150.0, tan_half_fov=float(p.sensor_width) * 0.5 / float(p.focal_length))
visits, kv = capi.make_visits(cols, visits_per_pixel=M, pixels_per_row=W)
ctx = capi.Context(0); ctx.set_params(p); ctx.set_lens(table); ctx.alloc_frame(1)
nbytes = n * 80
for it in range(3):
    t0 = time.perf_counter(); ctx.upload_visits(visits); ctx.sync(); t1 = time.perf_counter()
    ctx.clear_frame(); ctx.redistribute(); ctx.resolve(); ctx.sync(); t2 = time.perf_counter()
    print("upload %.1f ms = %.1f GB/s (pageable host memory) | pass %.2f ms | upload + pass %.1f Mvisits/s, pass alone %.0f Mvisits/s"
          % ((t1 - t0) * 1e3, nbytes / (t1 - t0) / 1e9, (t2 - t1) * 1e3, n / (t2 - t0) / 1e6, n / (t2 - t1) / 1e6))
