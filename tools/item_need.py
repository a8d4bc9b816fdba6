"""What an item needs: for every redistributed visit ("item") of a synthetic frame, how many of its attempts
succeed -- from the oracle's accepted-draw log (tests/, tools/ only: the oracle is the checker).

  python tools/item_need.py [--width 3840 --height 2160 --samples 1024 --seed 0x5EED --lens double_gauss_50mm]

Writes gpurun_out/item_need_<lens>_<seed>.npz: per item (px, py, x, y, z, samples, accepted, last_attempt) -- the data
the first-batch predictor of the streamed pass (lentil_hip.hip, `BatchModel`) was designed against.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from pota_amd import camera, capi, lens_io, workload  # noqa: E402
import common  # noqa: E402
import oracle_lib  # noqa: E402


def highlight_items(W, H, M, seed, f_hi, tan_half_fov, chunk_rows=60):
    out = {k: [] for k in ("rgba", "pos_z", "raydir_time", "volume_ignore", "transmission", "pixel")}
    for y0 in range(0, H, chunk_rows):
        y1 = min(H, y0 + chunk_rows)
        cols = workload.generate(np, y0 * W * M, y1 * W * M, W, H, M, seed=seed, f_hi=f_hi, focus_dist=150.0,
                                 tan_half_fov=tan_half_fov)
        hi = cols["rgba"][:, 0] > 2.0
        idx = np.nonzero(hi)[0]
        for k in ("rgba", "pos_z", "raydir_time", "volume_ignore", "transmission"):
            out[k].append(cols[k][idx])
        pix = (idx + y0 * W * M) // M
        out["pixel"].append(((pix % W) | ((pix // W) << 16)).astype(np.uint32))
    return {k: np.ascontiguousarray(np.concatenate(v, axis=0)) for k, v in out.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--samples", type=int, default=1024)
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED)
    ap.add_argument("--f-hi", type=float, default=2.0 ** -16)
    ap.add_argument("--lens", default="double_gauss_50mm")
    ap.add_argument("--threads", type=int, default=8)
    a = ap.parse_args()
    W, H, M = a.width, a.height, 9
    p, model, table, keep = common.po_setup(W, H, lens=a.lens, samples_override=a.samples)
    thf = common.tan_half_fov(p)
    cols = highlight_items(W, H, M, a.seed, a.f_hi, thf)
    n = cols["rgba"].shape[0]
    print("items (highlight visits):", n, file=sys.stderr)
    visits, kv = capi.make_visits(cols)
    lib = oracle_lib.load()
    fr = common.ThreadedOracle(lib, p, table, visits, a.threads, row_visits=max(1, n // (4 * a.threads)))
    log = fr.log()          # (visit, attempt, pixel)
    c = fr.counters()
    print("redistributed %d attempted %d accepted %d" % (c.redistributed_visits, c.attempted_draws, c.accepted_draws), file=sys.stderr)
    acc = np.bincount(log[:, 0], minlength=n)
    last = np.zeros(n, np.int64)
    np.maximum.at(last, log[:, 0], log[:, 1].astype(np.int64))
    px = (cols["pixel"] & 0xFFFF).astype(np.int32)
    py = (cols["pixel"] >> 16).astype(np.int32)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    f = os.path.join(ROOT, "gpurun_out", "item_need_%s_%x.npz" % (a.lens, a.seed))
    np.savez(f, px=px, py=py, pos=cols["pos_z"], accepted=acc, last_attempt=last, samples=a.samples, xres=p.xres, yres=p.yres)
    short = acc < a.samples
    print("items short of their draws (ran to 5 x samples): %d" % int(short.sum()), file=sys.stderr)
    need = np.where(short, 5 * a.samples, last + 1)
    more = need > a.samples + 16
    print("items needing more than samples + 16 attempts: %d; sum of needs %d vs %d" % (int(more.sum()), int(need.sum()), n * a.samples), file=sys.stderr)
    print(f)


if __name__ == "__main__":
    main()
