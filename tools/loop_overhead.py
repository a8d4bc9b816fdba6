"""Development aid: what the bench loop's own per-step reads (last_timing, counters, last_launches) cost the timed region.
usage: python3 tools/loop_overhead.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import bench
dev = torch.device("cuda:0")
b = bench.Bench(torch, None, dev, 0, 1, 0, 3840, 2160, 2160, 9, "double_gauss_50mm", 1024, 0, 2.0 ** -16, False)
b.generate(2.0 ** -16)
for _ in range(6):
    b.step()
for rep in range(3):
    for reads in (True, False):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(60):
            b.step()
            if reads:
                b.ctx.last_timing(); b.ctx.counters(); b.ctx.last_launches()
        b.ctx.sync(); torch.cuda.synchronize()
        print("reads" if reads else "no reads", "%.4f ms per step" % ((time.perf_counter() - t0) / 60 * 1e3))
