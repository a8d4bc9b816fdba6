#!/usr/bin/env python3
"""A highlight-heavy frame through the chunked form of the pass: the workload the solve kernels are profiled on
(rocprofv3 --pmc ... -- python3 tools/solve_workload.py <lens> [f_hi] [passes] [width height samples]).
Prints per pass: scan / draw ms, Newton lane-iterations per second, lane utilisation."""
import os, sys
os.environ.setdefault("LENTIL_STREAM", "0")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from pota_amd import camera, capi, lens_io, workload


def main():
    lens = sys.argv[1] if len(sys.argv) > 1 else "double_gauss_50mm"
    f_hi = float(sys.argv[2]) if len(sys.argv) > 2 else 1.6e-3
    passes = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    W = int(sys.argv[4]) if len(sys.argv) > 4 else 3840
    H = int(sys.argv[5]) if len(sys.argv) > 5 else 2160
    samples = int(sys.argv[6]) if len(sys.argv) > 6 else 1024
    M = 9
    p = camera.default_params()
    camera.setup_filter(p, W, H, filter_width=1.0, aa_samples=3)
    p, model = camera.setup_po(p, lens, focus_dist=150.0)
    p.samples_override = samples
    table, keep = lens_io.make_lens_table(model.spec)
    dev = torch.device("cuda:0")
    n = W * H * M
    cols = workload.generate(torch, 0, n, W, H, M, f_hi=f_hi, focus_dist=150.0,
                             tan_half_fov=float(p.sensor_width) * 0.5 / float(p.focal_length), n_extra=0, device=dev)
    torch.cuda.synchronize()
    ctx = capi.Context(0)
    ctx.set_params(p); ctx.set_lens(table); ctx.alloc_frame(1)
    visits, kv = capi.make_visits(cols, visits_per_pixel=M, pixels_per_row=W, ptr=lambda t: t.data_ptr())
    ctx.bind_visits(visits, kv)
    flops = sum(lens_io.newton_iteration_flops(model.spec))
    for it in range(passes):
        ctx.clear_frame(); ctx.redistribute(); ctx.resolve(); ctx.sync()
        ms = ctx.last_timing(); c = ctx.counters()
        print("pass %d: scan %.3f ms draw %.3f ms | items %d attempted %d accepted %d | tries %d iters %d (%.1f per try) "
              "lane-utilisation %.3f | %.2f G lane-iterations/s over the draws, %.1f TFLOP/s fp64 (%d flop per iteration)"
              % (it, ms[0], ms[1], c.redistributed_visits, c.attempted_draws, c.accepted_draws, c.tries, c.newton_iterations,
                 c.newton_iterations / max(c.tries, 1), c.newton_iterations / max(c.lane_rounds, 1),
                 c.newton_iterations / max(ms[1], 1e-9) / 1e6, c.newton_iterations * flops / max(ms[1], 1e-9) / 1e9, flops), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
