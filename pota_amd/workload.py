"""Seeded synthetic AOV-sample ("visit") streams -- SURVEY.md section 8(d).

The same integer recipe runs on numpy (host, parity tests) and on torch (device, bench): per visit a
tea<8>(visit, seed) hash seeds an LCG whose 24-bit outputs drive everything (the reference's own
generators, src/global.h:32-57).  Visit classes, colours and depths are bit-identical across the
two backends; the derived length / direction columns may differ in the last bit (torch's CPU sqrt
is not correctly rounded), which is why every comparison feeds oracle and GPU the very same arrays.

Frame W x H, M visits per pixel in pixel-major order; camera at the origin looking down -z,
identity world_to_camera, units cm, static.  Two visit classes:
  background : on the focus plane (+- small jitter), CoC < 0.4 -> not redistributed, RGB ~ U(0,1)
  highlight  : probability f_hi, out of focus (CoC well above 0.4), RGB = 84.1589 (the radiance in
               the reference's captured tests/cuda/sampledata.txt)
"""
import numpy as np

MASK = 0xFFFFFFFF
HIGHLIGHT_RADIANCE = 84.1589


def _tea8(xp, v0, v1):
    s0 = 0
    for _ in range(8):
        s0 = (s0 + 0x9E3779B9) & MASK
        v0 = (v0 + ((((v1 << 4) & MASK) + 0xA341316C) ^ (v1 + s0) ^ ((v1 >> 5) + 0xC8013EA4))) & MASK
        v1 = (v1 + ((((v0 << 4) & MASK) + 0xAD90777D) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7E95761E))) & MASK
    return v0


def _lcg(xp, s):
    s = (s * 1664525 + 1013904223) & MASK
    return s, (s & 0x00FFFFFF)


def generate(xp, v_begin, v_end, width, height, M, seed=0x5EED, f_hi=1.6e-3, focus_dist=150.0,
             tan_half_fov=0.36, n_extra=0, device=None, row_stride=1, row_offset=0):
    """Visits [v_begin, v_end) of the frame stream as a dict of [n,4] fp32 columns.

    With row_stride G and row_offset r only image rows y = r, r+G, ... are generated (the
    row-interleaved multi-GPU partition); visit ids then count within that sub-stream but the
    per-visit hash uses the *global* visit id, so the union over ranks equals the full stream."""
    is_torch = xp.__name__ == "torch"
    if is_torch:
        idx = xp.arange(v_begin, v_end, dtype=xp.int64, device=device)
        to_f = lambda a: a.to(xp.float32)
        stack = lambda cols: xp.stack(cols, dim=1).contiguous()
        zeros = lambda: xp.zeros(idx.shape[0], dtype=xp.float32, device=device)
    else:
        idx = xp.arange(v_begin, v_end, dtype=xp.int64)
        to_f = lambda a: a.astype(xp.float32)
        stack = lambda cols: xp.ascontiguousarray(xp.stack(cols, axis=1))
        zeros = lambda: xp.zeros(idx.shape[0], dtype=xp.float32)
    where, sqrt = xp.where, xp.sqrt
    # scalar constants are rounded to fp32 once, then used as (weak) python floats with fp32 arrays:
    # numpy (NEP 50) and torch both keep such expressions in fp32, so both backends agree bit for bit
    c = lambda x: float(np.float32(x))

    pix = idx // M
    px = pix % width
    py_local = pix // width
    py = py_local * row_stride + row_offset
    gvisit = (py * width + px) * M + (idx % M)          # global visit id of the full-frame stream

    s = _tea8(xp, gvisit & MASK, (gvisit * 0 + seed) & MASK)
    u = []
    for _ in range(8 + 4 * n_extra):
        s, r = _lcg(xp, s)
        u.append(to_f(r) * (1.0 / 16777216.0))

    # class: threshold on a 24-bit integer so both backends agree exactly
    thr = int(round(f_hi * 16777216.0))
    s, rclass = _lcg(xp, s)
    hi = rclass < thr

    fd = c(focus_dist)
    # depth (positive distance along -z), cm
    near = (u[0] * c(0.4) + c(0.35)) * fd
    far = (u[0] * c(2.4) + c(1.6)) * fd
    d_hi = where(u[1] < 0.5, near, far)
    d_bg = ((u[0] - 0.5) * c(0.002) + 1.0) * fd
    depth = where(hi, d_hi, d_bg)
    # position through the source pixel (sub-pixel jitter u2,u3), pinhole mapping
    fx = (to_f(px) + u[2]) * c(2.0 / width) - 1.0
    fy = 1.0 - (to_f(py) + u[3]) * c(2.0 / height)
    cx = c(tan_half_fov)
    cy = c(np.float32(tan_half_fov) * np.float32(height) / np.float32(width))
    X = fx * cx * depth
    Y = fy * cy * depth
    Zc = -depth
    dist = sqrt(X * X + Y * Y + Zc * Zc)     # last-bit backend dependent, see module docstring
    rad = c(HIGHLIGHT_RADIANCE)
    z0 = zeros()
    r_ = where(hi, z0 + rad, u[4])
    g_ = where(hi, z0 + rad, u[5])
    b_ = where(hi, z0 + rad, u[6])
    one = z0 + 1.0
    cols = {
        "rgba": stack([r_, g_, b_, one]),
        "pos_z": stack([X, Y, Zc, dist]),
        "raydir_time": stack([X / dist, Y / dist, Zc / dist, z0]),
        "volume_ignore": stack([z0, z0, z0, z0]),
        "transmission": stack([z0, z0, z0, z0]),
        "extra": [stack([u[8 + 4 * k], u[9 + 4 * k], u[10 + 4 * k], u[11 + 4 * k]]) for k in range(n_extra)],
    }
    return cols


def frame_visit_count(width, height, M, row_stride=1, row_offset=0):
    rows = len(range(row_offset, height, row_stride))
    return rows * width * M
