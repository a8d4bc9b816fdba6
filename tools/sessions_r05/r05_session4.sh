#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s4; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
PASSES=4 bash tools/pass_sequence.sh > $O/pass_sequence4.txt 2>&1
PASSES=2 LENTIL_PREDICT=0 bash tools/pass_sequence.sh > $O/pass_sequence_nopredict.txt 2>&1
python3 - > $O/per_pass.txt 2>&1 <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import bench
from pota_amd import capi
dev = torch.device("cuda:0")
b = bench.Bench(torch, None, dev, 0, 1, 0, 3840, 2160, 2160, 9, "double_gauss_50mm", 1024, 0, 2.0 ** -16, False)
b.generate(2.0 ** -16)
for k in range(16):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    b.step()
    b.ctx.sync(); dt = time.perf_counter() - t0
    c = b.ctx.counters()
    print(k, "ms %.3f" % (dt * 1e3), "timing", tuple(round(x, 3) for x in b.ctx.last_timing()), "tries", c.tries, "iters", c.newton_iterations, "parked", c.slow_solves,
          "rounds", b.ctx.last_launches()[1], "model", b.ctx.batch_model_stats())
PY
