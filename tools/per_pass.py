"""Development aid: the headline frame pass by pass (alternating streams): wall time, kernel times, solve counters, rounds,
first-batch-model state.  usage: python3 tools/per_pass.py [passes] [lens] [aovs]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
lens = sys.argv[2] if len(sys.argv) > 2 else "double_gauss_50mm"
aovs = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda:0")
b = bench.Bench(torch, None, dev, 0, 1, 0, 3840, 2160, 2160, 9, lens, 1024, aovs, 2.0 ** -16, False)
b.generate(2.0 ** -16)
ms = []
for k in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    b.step()
    b.ctx.sync(); dt = time.perf_counter() - t0
    c = b.ctx.counters()
    ms.append(dt * 1e3)
    print(k, "ms %.3f" % (dt * 1e3), "timing", tuple(round(x, 3) for x in b.ctx.last_timing()), "tries", c.tries, "iters", c.newton_iterations, "parked", c.slow_solves,
          "rounds", b.ctx.last_launches()[1], "model", b.ctx.batch_model_stats())
tail = sorted(ms[n // 2:])
print("median of the second half: %.3f ms, mean %.3f" % (tail[len(tail) // 2], sum(tail) / len(tail)))
