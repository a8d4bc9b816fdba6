"""BASELINE config 1 (thin lens, 512 x 512, 64 draws, beauty; the reference's own CPU-runnable case): the pass on the
GPU and the oracle on the host's cores, same visit stream.  Prints one JSON object."""
import json
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
import common
import oracle_lib
from pota_amd import capi

W = H = 512
M = 9
p = common.tl_setup(W, H, samples_override=64)
visits, cols = common.make_stream(p, W, H, M, f_hi=2.0 ** -9)
n = int(visits.n)
orc = oracle_lib.load()
t0 = time.perf_counter()
ref = common.run_oracle(orc, p, None, visits, keep_log=False)
t_cpu = time.perf_counter() - t0
rc = ref.counters()
ctx = capi.Context(0)
ctx.set_params(p); ctx.set_bokeh(None); ctx.alloc_frame(1)
ctx.upload_visits(visits)
for _ in range(3):
    ctx.clear_frame(); ctx.redistribute(); ctx.resolve()
ctx.sync()
K = 20
t0 = time.perf_counter()
for _ in range(K):
    ctx.clear_frame(); ctx.redistribute(); ctx.resolve()
ctx.sync()
t_gpu = (time.perf_counter() - t0) / K
c = ctx.counters()
img, rimg = ctx.download_aov(0), ref.resolve(0)
m = rimg != 0
out = {"config": "1: thin lens 512x512, 64 draws, beauty, f_hi=2^-9", "visits": n,
       "redistributed": int(rc.redistributed_visits), "attempted": int(rc.attempted_draws), "accepted": int(rc.accepted_draws),
       "counters_equal": (int(c.redistributed_visits), int(c.attempted_draws), int(c.accepted_draws)) ==
                         (int(rc.redistributed_visits), int(rc.attempted_draws), int(rc.accepted_draws)),
       "max_rel_err_resolved": float(np.max(np.abs(img[m] - rimg[m]) / np.abs(rimg[m]))),
       "cpu_oracle_1_thread_Mvisits_s": round(n / t_cpu / 1e6, 3), "cpu_s": round(t_cpu, 2),
       "gpu_ms_per_pass": round(t_gpu * 1e3, 4), "gpu_Mvisits_s": round(n / t_gpu / 1e6, 1),
       "gpu_Mdraws_s": round(int(c.attempted_draws) / t_gpu / 1e6, 1)}
print(json.dumps(out))
