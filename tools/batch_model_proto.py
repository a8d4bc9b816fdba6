"""Prototype (numpy + the oracle; tools/ only) of the first-batch predictor of the streamed pass.

Calibration: on a grid of camera-space targets (field x, field y, inverse depth) trace K stratified aperture points through
the lens (the oracle's lens_lt_sample_aperture) and keep, per node and point, where the ray lands in pixel coordinates or
that it is vignetted.  Prediction for an item: trilinear blend of the eight nodes around its target, per point; the share
of points that land inside the frame is the success rate of its attempts, and the first batch is sized from it.

  python tools/batch_model_proto.py gpurun_out/item_need_double_gauss_50mm_5eed.npz [...]

Prints, per data file, how the predicted batch compares with what the oracle says every item needed.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import common  # noqa: E402
import oracle_lib  # noqa: E402


def disk_points(lib, n_side):
    pts = []
    out = (C.c_double * 2)()
    for i in range(n_side):
        for j in range(n_side):
            d1, d2 = (i + 0.5) / n_side, (j + 0.5) / n_side
            lib.orc_concentric_disk_sample(d2, d1, out)
            pts.append((out[0], out[1]))
    return np.array(pts)


class Model:
    def __init__(self, lib, p, table, nx=17, ny=11, nz=9, n_side=16, field_margin=1.0, u_max=None):
        self.lib, self.p = lib, p
        self.lens = lib.orc_lens_create(C.byref(table))
        thx = float(p.sensor_width) * 0.5 / float(p.focal_length)
        thy = thx * float(p.yres_without_region) / float(p.xres_without_region)
        self.fx = np.linspace(-thx * field_margin, thx * field_margin, nx)
        self.fy = np.linspace(-thy * field_margin, thy * field_margin, ny)
        fd_cm = float(p.focus_distance) / 10.0
        if u_max is None:
            u_max = 4.0 / fd_cm
        self.u = np.linspace(1e-6, u_max, nz)
        ap = disk_points(lib, n_side) * float(p.aperture_radius)
        self.K = ap.shape[0]
        self.land = np.full((nz, ny, nx, self.K, 2), np.nan)
        lam = float(p.lambda_bw)
        sensor = (C.c_double * 5)()
        out = (C.c_double * 5)()
        tgt = (C.c_double * 3)()
        apc = (C.c_double * 2)()
        k = table
        bfl, ipr = float(k.lens_back_focal_length), float(k.lens_inner_pupil_radius)
        sw, shift = float(p.sensor_width), float(p.sensor_shift)
        aspect = float(p.xres_without_region) / float(p.yres_without_region)
        for iz, u in enumerate(self.u):
            d = 1.0 / u
            for iy, fy in enumerate(self.fy):
                for ix, fx in enumerate(self.fx):
                    # camera space (cm): (fx d, fy d, -d); target = -P_cs * 10 (src/lentil_filter.cpp:271)
                    tgt[0], tgt[1], tgt[2] = -fx * d * 10.0, -fy * d * 10.0, d * 10.0
                    for kk in range(self.K):
                        apc[0], apc[1] = ap[kk]
                        for i in range(4):
                            sensor[i] = 0.0
                            out[i] = 0.0
                        sensor[4] = lam; out[4] = lam
                        t = lib.orc_lt_sample_aperture(self.lens, tgt, apc, sensor, out, lam, None)
                        if not (np.float32(t) > 0):
                            continue
                        ipx = sensor[0] + sensor[2] * bfl
                        ipy = sensor[1] + sensor[3] * bfl
                        if ipx * ipx + ipy * ipy > ipr * ipr:
                            continue
                        sx = sensor[0] + sensor[2] * -shift
                        sy = sensor[1] + sensor[3] * -shift
                        s0 = sx / (sw * 0.5)
                        s1 = sy / (sw * 0.5) * aspect
                        self.land[iz, iy, ix, kk, 0] = ((s0 + 1.0) / 2.0) * p.xres_without_region - p.region_min_x
                        self.land[iz, iy, ix, kk, 1] = ((-s1 + 1.0) / 2.0) * p.yres_without_region - p.region_min_y

    def rate(self, x, y, z, border=0.0):
        """(share of the K points that land inside the frame, share vignetted) for a camera-space point"""
        d = -z
        fx, fy, u = x / d, y / d, 1.0 / d

        def cell(a, v):
            t = (v - a[0]) / (a[1] - a[0])
            t = min(max(t, 0.0), len(a) - 1.0)
            i = min(int(t), len(a) - 2)
            return i, t - i
        ix, tx = cell(self.fx, fx)
        iy, ty = cell(self.fy, fy)
        iz, tz = cell(self.u, u)
        acc = np.zeros((self.K, 2))
        wgood = np.zeros(self.K)
        for dz, wz in ((0, 1 - tz), (1, tz)):
            for dy, wy in ((0, 1 - ty), (1, ty)):
                for dx, wx in ((0, 1 - tx), (1, tx)):
                    L = self.land[iz + dz, iy + dy, ix + dx]
                    ok = ~np.isnan(L[:, 0])
                    w = wz * wy * wx
                    wgood += ok * w
                    acc += np.nan_to_num(L) * w
        # a point counts as far as the nodes around let it through; it lands where those nodes say
        have = wgood > 1e-6
        acc = acc / np.maximum(wgood, 1e-9)[:, None]
        p = self.p

        def inside(border):
            return have & (acc[:, 0] >= border) & (acc[:, 0] < p.xres - border) & (acc[:, 1] >= border) & (acc[:, 1] < p.yres - border)
        tot = wgood.sum()
        fail = 1.0 - tot / self.K
        if tot <= 0:
            return 0.0, 0.0, 1.0
        return float((wgood * inside(border)).sum() / tot), float((wgood * inside(0.0)).sum() / tot), fail


def first_batch(S, q_strict, q, fail, retries, rel=0.08, abs_=0.02, sigmas=4.0, spare=8):
    limit = 5 * S + retries
    allfail = fail ** (retries + 1)
    if q_strict >= 1.0:
        qa = 1.0 - allfail
        if allfail < 1e-7:
            return min(limit, S + retries + spare)
    else:
        qa = max(0.0, q * (1.0 - rel) - abs_) * (1.0 - allfail)
    if qa <= 0.02:
        return limit
    n = S / qa + sigmas * np.sqrt(S * (1.0 - qa)) / qa + retries + spare
    return int(min(limit, np.ceil(n)))


def main():
    lib = oracle_lib.load()
    files = sys.argv[1:]
    d0 = np.load(files[0])
    lens = "double_gauss_50mm"
    for f in files:
        if "petzval" in f:
            lens = "petzval_58mm"
    W, H = int(d0["xres"]) - 1, int(d0["yres"]) - 1
    S = int(d0["samples"])
    p, model, table, keep = common.po_setup(W, H, lens=lens, samples_override=S)
    import time
    t0 = time.time()
    M = Model(lib, p, table)
    print("calibration: %d solves, %.1f s" % (M.land.size // 2, time.time() - t0), file=sys.stderr)
    retries = int(p.vignetting_retries)
    for f in files:
        d = np.load(f)
        pos, acc, last = d["pos"], d["accepted"], d["last_attempt"]
        need = np.where(acc < S, 5 * S, last + 1) + 0
        n = len(need)
        batch = np.zeros(n, np.int64)
        qs = np.zeros(n)
        for i in range(n):
            qst, q, fail = M.rate(float(pos[i, 0]), float(pos[i, 1]), float(pos[i, 2]), border=3.0)
            qs[i] = q
            batch[i] = first_batch(S, qst, q, fail, retries)
        # an item is served when its batch covers every R(m) it looks at: attempts [0, need) and their retries
        short = (batch < np.minimum(need + retries, 5 * S + retries)) & (need > 0)
        base = n * (S + retries + 16)
        print('   items with a batch above the plain one:', int((batch > S + retries + 8).sum()))
        print("%s: items %d, short %d, batch total %d vs base %d (+%.2f %%), true need %d"
              % (os.path.basename(f), n, int(short.sum()), int(batch.sum()), base, 100.0 * (batch.sum() - base) / base, int(need.sum())))
        true_q = np.where(acc < S, acc / (5.0 * S), S / np.maximum(1, last + 1.0))
        o = np.argsort(true_q)
        for i in o[:40]:
            print("   item %4d  true q %.3f  model qa %.3f  need %5d  batch %5d %s" % (i, true_q[i], qs[i], need[i], batch[i], "SHORT" if short[i] else ""))


if __name__ == "__main__":
    main()
