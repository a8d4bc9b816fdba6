"""Development aid: host time stamps of the bench loop's calls on the headline frame (LENTIL_HOST_TRACE): where a step's time
goes between the device's last kernel and the next pass's first launch.  usage: python3 tools/host_trace.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("LENTIL_HOST_TRACE", "40")
import torch
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
b = bench.Bench(torch, None, dev, 0, 1, 0, 3840, 2160, 2160, 9, "double_gauss_50mm", 1024, 0, 2.0 ** -16, False)
b.generate(2.0 ** -16)
for _ in range(8):
    b.step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n):
    b.step()
b.ctx.sync(); torch.cuda.synchronize()
print("%.4f ms per step" % ((time.perf_counter() - t0) / n * 1e3))
