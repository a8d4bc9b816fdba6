#!/usr/bin/env python3
"""What the kernels of ONE streamed pass do when (development aid).

    python tools/timeline.py --build          # here: hipcc -DLENTIL_TIMELINE -> pota_amd/_ab/liblentil_hip_tl.so
    python tools/timeline.py [--passes 5]     # on the GPU box: bench.py's headline workload, alternating streams

The instrumented build counts events into 20.48 us buckets of the chip-wide 100 MHz counter (lentil_kernels.h, tl_add):
scan tiles done, tasks published, Newton lane-iterations of the solve instances, parked-solve iterations, items accepted,
empty polls of the task queue.  Printed per bucket for the last pass, plus the backlog (tasks published - taken).
"""
import argparse
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
SO = os.path.join(ROOT, "pota_amd", "_ab", "liblentil_hip_tl.so")
NAMES = ["tiles", "tasks_pub", "itersA", "itersB", "itersR2+", "slow_it", "items_acc", "polls0", "tasks_taken", "parked"]


def build():
    import __graft_entry__ as g
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    extra = os.environ.get("LENTIL_TL_FLAGS", "").split()      # e.g. -DLENTIL_ACCEPT_EU=4
    g.hip_variant(SO, defines=[f for f in extra if f.startswith("-D")] + ["-DLENTIL_TIMELINE"], extra=[f for f in extra if not f.startswith("-D")])
    print("built", SO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--passes", type=int, default=5)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--samples", type=int, default=1024)
    ap.add_argument("--f-hi", type=float, default=2.0 ** -16)
    ap.add_argument("--lens", default="double_gauss_50mm")
    ap.add_argument("--aovs", type=int, default=0, help="extra AOVs (BASELINE config 4: --lens petzval_58mm --aovs 8)")
    ap.add_argument("--log", type=int, default=0, help="draw log capacity (records)")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    if args.build:
        return build()
    os.environ["LENTIL_HIP_LIB"] = SO
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import numpy as np
    import torch
    import common
    from pota_amd import capi, workload
    lib = capi.load_library()
    lib.lentil_hip_debug_timeline.restype = C.c_int
    lib.lentil_hip_debug_timeline.argtypes = [C.c_void_p, C.c_int]
    W, H, M = args.width, args.height, 9
    p, model, table, keep = common.po_setup(W, H, lens=args.lens, samples_override=args.samples)
    dev = torch.device("cuda:0")
    ctx = capi.Context(0)
    ctx.set_params(p); ctx.set_lens(table); ctx.set_bokeh(None); ctx.alloc_frame(1 + args.aovs); ctx.set_draw_log(args.log)
    streams = []
    for seed in (0x5EED, 0xBEEF):
        cols = workload.generate(torch, 0, W * H * M, W, H, M, seed=seed, f_hi=args.f_hi, focus_dist=150.0,
                                 tan_half_fov=common.tan_half_fov(p), n_extra=args.aovs, device=dev)
        streams.append((cols, capi.make_visits(cols, visits_per_pixel=M, pixels_per_row=W, ptr=lambda t: t.data_ptr())))
    torch.cuda.synchronize()
    import time
    for i in range(args.passes):
        cols, (v, kv) = streams[i % 2]
        ctx.bind_visits(v, kv)
        if i == args.passes - 1:
            ctx.sync()
            lib.lentil_hip_debug_timeline(None, 1)
        t0 = time.perf_counter()
        try:
            ctx.clear_frame(); ctx.redistribute(); ctx.resolve(); ctx.sync()
        except capi.LentilError as e:
            print("pass %d: %s" % (i, e))
            if i == args.passes - 1:
                raise
        dt = time.perf_counter() - t0
    c = ctx.counters()
    raw = np.zeros(64 * 10 * 1024 + 64 + 128, np.uint32)
    lib.lentil_hip_debug_timeline(raw.ctypes.data, 0)
    h = raw[:64 * 10 * 1024].reshape(64, 10, 1024).sum(axis=0)
    dbg = raw[64 * 10 * 1024:64 * 10 * 1024 + 64].view(np.uint64)
    span = raw[64 * 10 * 1024 + 64:].view(np.uint64).reshape(32, 2)
    tot = h.sum(axis=0)
    used = np.nonzero(tot)[0]
    lines = ["last pass: %.3f ms host time, streamed %d, timing ms %s" % (dt * 1e3, c.streamed, ctx.last_timing())]
    if used.size:
        # the window may wrap around the 1024-bucket ring
        gaps = np.diff(np.concatenate([used, [used[0] + 1024]]))
        start = int(used[(int(np.argmax(gaps)) + 1) % used.size])
        order = [(start + k) % 1024 for k in range(1024)]
        last = max(k for k, b in enumerate(order) if tot[b])
        lines.append("%7s " % "t_us" + " ".join("%9s" % n for n in NAMES) + "   backlog_tasks")
        pub = taken = 0
        for k in range(last + 1):
            b = order[k]
            pub += int(h[1, b]); taken += int(h[8, b])
            lines.append("%7.0f " % (k * 20.48) + " ".join("%9d" % int(h[ch, b]) for ch in range(10)) + "   %d" % (pub - taken))
        lines.append("totals  " + " ".join("%9d" % int(h[ch].sum()) for ch in range(10)))
    SPANS = ["scan", "publish", "solve r0 (A)", "solve r1", "solve r2+", "slow r0", "slow r1", "slow r2+", "accept<0>", "accept<1>",
             "accept<2>", "resolve", "resolve_touched", "clear_touched", "reset_round", "prep"]
    live = [(int(span[i, 0]), int(span[i, 1]), SPANS[i]) for i in range(len(SPANS)) if span[i, 1] > 0]
    if live:
        t0 = min(s for s, e, n in live)
        lines.append("kernel spans (us from the first kernel's first block; first block in .. last block out):")
        for s_, e_, n in sorted(live):
            lines.append("  %-16s %8.1f .. %8.1f  (%7.1f)" % (n, (s_ - t0) / 100.0, (e_ - t0) / 100.0, (e_ - s_) / 100.0))
        lines.append("  host: redistribute+resolve returned after %.1f us" % (dt * 1e6))
    lines.append("dbg: stragglers pixel/out/fail %d/%d/%d, ended <100 / at 100 iterations %d/%d" % tuple(int(x) for x in dbg[:5]))
    lines.append("dbg: first accept, items by unknown attempts 0 / 1-4 / 5-16 / 17-64 / 65-256 / >256: %s; unknown attempts %d; items with unknowns scheduling more %d"
                 % ([int(x) for x in dbg[8:14]], int(dbg[14]), int(dbg[15])))
    lines.append("dbg: second accept, replayed items %d, of them still short %d, accepted %d of %d" % tuple(int(x) for x in dbg[16:20]))
    if int(dbg[27]):
        k = float(dbg[27]) * 100.0
        lines.append("dbg: accept_item_wide, us per item (thread 0's clock): metadata %.1f | wait for the block %.1f, window %.1f, passes 1-2 + prefix %.1f, "
                     "tops %.1f, decision (thread 0) %.1f, splats %.1f  (%d items)"
                     % (dbg[26] / k, dbg[20] / k, dbg[21] / k, dbg[22] / k, dbg[23] / k, dbg[24] / k, dbg[25] / k, int(dbg[27])))
    txt = "\n".join(lines)
    print(txt)
    if args.out:
        with open(args.out, "w") as f:
            f.write(txt + "\n")
    ctx.close()


if __name__ == "__main__":
    main()
