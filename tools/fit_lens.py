#!/usr/bin/env python3
"""Build a polynomial-optics lens table from a lens prescription.

The reference (zpelgrims/pota) does not ship polynomial coefficients: its
include/auto_generated_lens_includes/*.h only #include generated code from a
sibling checkout of zpelgrims/polynomial-optics that is absent (SURVEY.md §0.1,
Appendix D).  This script is the build's own stand-in for that generator: a
sequential spherical-surface ray tracer plus a sparse least-squares fit of
5-variate monomials (orthogonal matching pursuit), which is what the upstream
generator does in spirit.  It produces

  P_out : sensor (x, y, dx, dy, lambda) -> outer pupil (x, y, dx, dy) + transmittance
  P_ap  : sensor (x, y, dx, dy, lambda) -> aperture plane (x, y, dx, dy)

in the conventions the reference's call sites expect (src/lentil.h:1257-1313):
lengths in mm, lambda in micrometres, +z from the sensor towards the scene,
outer-pupil position/direction in the sphere parametrisation of
src/lens.h:99-153 (sphereToCs / csToSphere with centre -R, radius R).

Usage: python tools/fit_lens.py [lens-name ...]   (writes pota_amd/lenses/<name>.json)
"""
import json
import math
import os
import sys

import numpy as np

# ---------------------------------------------------------------------------
# Prescriptions, listed scene -> sensor (standard optical-design order):
# (radius, thickness to next surface, nd, vd, semi-diameter).  radius 0 = flat.
# nd = 1 means air.  'stop' marks the aperture plane.
# ---------------------------------------------------------------------------
AIR = (1.0, 0.0)
SK2 = (1.60738, 56.65)
SK16 = (1.62041, 60.32)
F5 = (1.60342, 38.03)
BK7 = (1.5168, 64.17)
F2 = (1.62004, 36.37)
SF2 = (1.64769, 33.85)
K5 = (1.52249, 59.48)

# Classic six-element double Gauss (the widely reproduced 100 mm f/3 textbook
# design), scaled by 0.5 to a 50 mm-class lens.
_DG = [
    (54.153, 8.747, SK2, 29.225),
    (152.522, 0.5, AIR, 28.141),
    (35.951, 14.0, SK16, 24.295),
    (0.0, 3.777, F5, 21.297),
    (22.270, 14.253, AIR, 14.919),
    ("stop", 12.428, AIR, 10.229),
    (-25.685, 3.777, F5, 13.188),
    (0.0, 10.834, SK16, 16.468),
    (-36.980, 0.5, AIR, 18.930),
    (196.417, 6.858, SK16, 21.311),
    (-67.148, 57.315, AIR, 21.646),
]

# Petzval-class portrait lens: two separated cemented achromats, fast and
# deliberately under-corrected for field curvature (the "swirl" look), 58 mm.
_PZ = [
    (55.9, 7.2, BK7, 20.5),
    (-43.7, 2.2, F2, 20.5),
    (460.4, 22.0, AIR, 20.0),
    ("stop", 18.0, AIR, 19.0),
    (110.6, 2.0, SF2, 17.0),
    (38.9, 3.2, AIR, 17.0),
    (48.0, 6.5, K5, 18.0),
    (-157.8, 40.7, AIR, 18.0),
]

# The same Petzval behind a weak cylindrical front attachment (curved in x only: a 1.33x-style anamorphic squeeze):
# the outer pupil is then a cylinder with its axis along y -- lens_outer_pupil_geometry "cyl-y", the
# cylinderToCs / csToCylinder parametrisation of src/lens.h:156-221.  A sixth entry "cyl-y" marks such a surface.
_PZ_ANA = [
    (120.0, 3.5, BK7, 23.0, "cyl-y"),
    (0.0, 6.0, AIR, 23.0),
] + _PZ

LENSES = {
    # name: (prescription, scale, max_degree, n_terms, sensor half-extent used for the fit [mm])
    "double_gauss_50mm": dict(rx=_DG, scale=0.5, degree=7, terms=36, field=18.0),
    "petzval_58mm": dict(rx=_PZ, scale=0.4745, degree=9, terms=48, field=16.0, refocus=True),
    "anamorphic_petzval_58mm": dict(rx=_PZ_ANA, scale=0.4745, degree=7, terms=32, field=14.0, refocus=True, outer="cyl-y"),
}


def cauchy(nd, vd, lam):
    """Two-term Cauchy dispersion through (nd, vd); lam in micrometres."""
    if nd == 1.0:
        return np.ones_like(lam)
    lF, lC, ld = 0.4861, 0.6563, 0.5876
    B = (nd - 1.0) / vd / (1.0 / lF ** 2 - 1.0 / lC ** 2)
    A = nd - B / ld ** 2
    return A + B / lam ** 2


def paraxial_bfl(rx, lam=0.5876):
    """Back focal distance (last surface -> paraxial focus of an axial object at infinity)."""
    y, nu, n = 1.0, 0.0, 1.0          # ray height, n*u, current index
    for i, (R, t, glass, sd) in enumerate(rx):
        n2 = float(cauchy(*glass, np.array([lam]))[0])
        if R != "stop" and R != 0.0:
            nu = nu - y * (n2 - n) / R
        n = n2 if R != "stop" else n
        if i < len(rx) - 1:
            y = y + (nu / n) * t
    return -y / (nu / n)


class Lens:
    def __init__(self, rx, scale, refocus=False):
        if refocus:     # put the sensor at the paraxial focus for infinity
            rx = list(rx)
            rx[-1] = (rx[-1][0], paraxial_bfl([e[:4] for e in rx if not (len(e) > 4)]), rx[-1][2], rx[-1][3])
        self.rx = rx
        self.surf = []
        z = 0.0
        for ent in rx:
            R, t, glass, sd = ent[:4]
            stop = R == "stop"
            self.surf.append(dict(R=0.0 if stop else R * scale, z=z, glass=glass,
                                  sd=sd * scale, stop=stop, cyl=(len(ent) > 4 and ent[4] == "cyl-y")))
            z += t * scale
        self.z_sensor = z            # standard frame: surface 1 vertex at z=0, sensor at +z
        self.bfl = self.rx[-1][1] * scale
        self.i_stop = [i for i, s in enumerate(self.surf) if s["stop"]][0]

    def trace(self, x, y, dx, dy, lam, to_aperture=False):
        """Trace from the sensor towards the scene.  Inputs are lentil-frame
        (direction (dx,dy,1)); works internally in the standard frame (z -> -z)."""
        n = x.shape[0]
        o = np.stack([x, y, np.full(n, self.z_sensor)], 1)
        d = np.stack([dx, dy, -np.ones(n)], 1)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        ok = np.ones(n, bool)
        T = np.ones(n)
        ap = None
        for i in range(len(self.surf) - 1, -1, -1):
            s = self.surf[i]
            R = s["R"]
            if R == 0.0:
                t = (s["z"] - o[:, 2]) / d[:, 2]
                p = o + t[:, None] * d
                nrm = np.tile(np.array([0.0, 0.0, 1.0]), (n, 1))
            elif s.get("cyl"):
                # cylinder with its axis along y through (0, *, z + R): x^2 + (z - c)^2 = R^2
                c = np.array([0.0, 0.0, s["z"] + R])
                oc = o - c
                a2 = d[:, 0] ** 2 + d[:, 2] ** 2
                b = (oc[:, 0] * d[:, 0] + oc[:, 2] * d[:, 2]) / a2
                cc = (oc[:, 0] ** 2 + oc[:, 2] ** 2 - R * R) / a2
                disc = b * b - cc
                ok &= disc > 0
                sq = np.sqrt(np.maximum(disc, 0.0))
                t1, t2 = -b - sq, -b + sq
                p1 = o + t1[:, None] * d
                use1 = np.sign(p1[:, 2] - c[2]) == -np.sign(R)
                t = np.where(use1, t1, t2)
                p = o + t[:, None] * d
                nrm = -np.stack([p[:, 0] - c[0], np.zeros(n), p[:, 2] - c[2]], 1) / R
            else:
                c = np.array([0.0, 0.0, s["z"] + R])
                oc = o - c
                b = np.einsum("ij,ij->i", oc, d)
                cc = np.einsum("ij,ij->i", oc, oc) - R * R
                disc = b * b - cc
                ok &= disc > 0
                sq = np.sqrt(np.maximum(disc, 0.0))
                # the vertex-side root: hit point has sign(p.z - c.z) == -sign(R)
                t1, t2 = -b - sq, -b + sq
                p1 = o + t1[:, None] * d
                use1 = np.sign(p1[:, 2] - c[2]) == -np.sign(R)
                t = np.where(use1, t1, t2)
                p = o + t[:, None] * d
                nrm = -(p - c) / R           # oriented against the (-z travelling) ray
            ok &= (p[:, 0] ** 2 + p[:, 1] ** 2) <= s["sd"] ** 2
            if s["stop"]:
                ap = np.stack([p[:, 0], p[:, 1], d[:, 0] / -d[:, 2], d[:, 1] / -d[:, 2]], 1)
                o = p
                if to_aperture:
                    return ap, ok
                continue
            n1 = cauchy(*s["glass"], lam)                                    # image-side medium
            n2 = cauchy(*(self.surf[i - 1]["glass"] if i > 0 else AIR), lam)  # object-side medium
            cosi = -np.einsum("ij,ij->i", d, nrm)
            ok &= cosi > 0
            eta = n1 / n2
            k = 1.0 - eta * eta * (1.0 - cosi * cosi)
            ok &= k > 0
            cost = np.sqrt(np.maximum(k, 0.0))
            d = eta[:, None] * d + (eta * cosi - cost)[:, None] * nrm
            d /= np.linalg.norm(d, axis=1, keepdims=True)
            rs = ((n1 * cosi - n2 * cost) / (n1 * cosi + n2 * cost)) ** 2
            rp = ((n1 * cost - n2 * cosi) / (n1 * cost + n2 * cosi)) ** 2
            T *= 1.0 - 0.5 * (rs + rp)
            o = p
        # outer pupil, lentil frame + sphere parametrisation (src/lens.h:127-153)
        R1 = self.surf[0]["R"]
        pos = np.stack([o[:, 0], o[:, 1], -o[:, 2]], 1)
        dr = np.stack([d[:, 0], d[:, 1], -d[:, 2]], 1)
        nz = np.abs((pos[:, 2] + R1) / R1)
        if self.surf[0].get("cyl"):     # csToCylinder, cyl-y (src/lens.h:190-221): the normal has no y component
            nrm = np.stack([pos[:, 0] / R1, np.zeros(n), nz], 1)
        else:
            nrm = np.stack([pos[:, 0] / R1, pos[:, 1] / R1, nz], 1)
        ex = np.stack([nrm[:, 2], np.zeros(n), -nrm[:, 0]], 1)
        ex /= np.linalg.norm(ex, axis=1, keepdims=True)
        ey = np.cross(nrm, ex)
        out = np.stack([pos[:, 0], pos[:, 1],
                        np.einsum("ij,ij->i", dr, ex), np.einsum("ij,ij->i", dr, ey)], 1)
        return out, ap, T, ok


def monomials(max_degree, lam_degree=2):
    """All exponent tuples (ex, ey, edx, edy, el) with spatial degree <= max_degree."""
    out = []
    for a in range(max_degree + 1):
        for b in range(max_degree + 1 - a):
            for c in range(max_degree + 1 - a - b):
                for d in range(max_degree + 1 - a - b - c):
                    for l in range(lam_degree + 1):
                        out.append((a, b, c, d, l))
    return out


def parity_ok(e, kind):
    """Symmetry of a rotationally symmetric system about the x/y mirror planes."""
    a, b, c, d, _ = e
    if kind == "x":     # odd in (x,dx), even in (y,dy)
        return (a + c) % 2 == 1 and (b + d) % 2 == 0
    if kind == "y":
        return (a + c) % 2 == 0 and (b + d) % 2 == 1
    return (a + c) % 2 == 0 and (b + d) % 2 == 0    # scalar (transmittance)


def omp_fit(X, y, exps, n_terms):
    """Orthogonal matching pursuit on column-normalised design matrix."""
    norms = np.linalg.norm(X, axis=0)
    norms[norms == 0] = 1.0
    Xn = X / norms
    sel = []
    resid = y.copy()
    coef = None
    for _ in range(min(n_terms, X.shape[1])):
        corr = np.abs(Xn.T @ resid)
        corr[sel] = -1.0
        j = int(np.argmax(corr))
        sel.append(j)
        coef, *_ = np.linalg.lstsq(Xn[:, sel], y, rcond=None)
        resid = y - Xn[:, sel] @ coef
    c = coef / norms[sel]
    # stable print order: ascending total degree, then lexicographic (like generated code)
    order = sorted(range(len(sel)), key=lambda k: (sum(exps[sel[k]][:4]), exps[sel[k]]))
    terms = [(float(c[k]), list(exps[sel[k]])) for k in order]
    rms = float(np.sqrt(np.mean(resid ** 2)))
    return terms, rms


def build(name, spec, seed=1234, n_rays=400000, n_fit=40000):
    lens = Lens(spec["rx"], spec["scale"], refocus=spec.get("refocus", False))
    rng = np.random.default_rng(seed)
    F = spec["field"]
    x = rng.uniform(-F, F, n_rays)
    y = rng.uniform(-F, F, n_rays)
    last = lens.surf[-1]
    # aim at a disk on the rear element (inner pupil), slightly over-filled
    r = last["sd"] * 1.05 * np.sqrt(rng.uniform(0, 1, n_rays))
    ph = rng.uniform(0, 2 * np.pi, n_rays)
    dx = (r * np.cos(ph) - x) / lens.bfl
    dy = (r * np.sin(ph) - y) / lens.bfl
    lam = rng.uniform(0.40, 0.70, n_rays)
    out, ap, T, ok = lens.trace(x, y, dx, dy, lam)
    idx = np.nonzero(ok)[0][:n_fit]
    print(f"[{name}] rays passing: {ok.mean():.3f}; fitting on {idx.size}")
    V = np.stack([x, y, dx, dy, lam], 1)[idx]
    exps_all = monomials(spec["degree"])

    def design(exps):
        cols = []
        for e in exps:
            col = np.ones(V.shape[0])
            for v in range(5):
                if e[v]:
                    col = col * V[:, v] ** e[v]
            cols.append(col)
        return np.stack(cols, 1)

    polys = {}
    targets = [("out_x", out[idx, 0], "x"), ("out_y", out[idx, 1], "y"),
               ("out_dx", out[idx, 2], "x"), ("out_dy", out[idx, 3], "y"),
               ("out_t", T[idx], "s"),
               ("ap_x", ap[idx, 0], "x"), ("ap_y", ap[idx, 1], "y"),
               ("ap_dx", ap[idx, 2], "x"), ("ap_dy", ap[idx, 3], "y")]
    cache = {}
    for pname, tgt, kind in targets:
        if kind not in cache:
            ex = [e for e in exps_all if parity_ok(e, kind)]
            cache[kind] = (ex, design(ex))
        ex, X = cache[kind]
        nt = spec["terms"] if pname != "out_t" else max(12, spec["terms"] // 2)
        terms, rms = omp_fit(X, tgt, ex, nt)
        polys[pname] = terms
        print(f"  {pname:7s} {len(terms):3d} terms  rms residual {rms:.3e}")

    # ----- lens constants (field names of src/lentil.h:106-120) -----
    first = lens.surf[0]
    stop = lens.surf[lens.i_stop]
    # paraxial effective focal length: trace a near-axis parallel bundle backwards is awkward;
    # use the marginal-ray estimate from the sensor side instead.
    h = 1e-3
    o2, _, _, _ = lens.trace(np.array([0.0]), np.array([0.0]), np.array([h]), np.array([0.0]),
                             np.array([0.55]))
    # a ray leaving the on-axis sensor point with slope h exits (focused at infinity) parallel
    # at height ~ f*h
    efl = abs(o2[0, 0] / h)
    # wide-open f-number from the stop: exit-pupil marginal slope at the sensor
    a2, okk = lens.trace(np.zeros(2001), np.zeros(2001), np.linspace(0, 0.6, 2001),
                         np.zeros(2001), np.full(2001, 0.55), to_aperture=True)
    umax = np.linspace(0, 0.6, 2001)[okk][-1]
    fstop = 1.0 / (2.0 * math.sin(math.atan(umax)))
    consts = dict(
        lens_outer_pupil_radius=first["sd"],
        lens_inner_pupil_radius=last["sd"],
        lens_length=lens.z_sensor,
        lens_back_focal_length=lens.bfl,
        lens_effective_focal_length=efl,
        lens_aperture_pos=lens.z_sensor - stop["z"],
        lens_aperture_housing_radius=stop["sd"],
        lens_inner_pupil_curvature_radius=-last["R"] if last["R"] else 1e6,
        lens_outer_pupil_curvature_radius=first["R"],
        lens_field_of_view=2.0 * math.atan(F / efl),
        lens_fstop=fstop,
        lens_aperture_radius_at_fstop=stop["sd"],
        lens_inner_pupil_geometry="spherical",
        lens_outer_pupil_geometry=spec.get("outer", "spherical"),
    )
    table = dict(name=name, constants=consts, polys=polys,
                 note="self-fitted by tools/fit_lens.py; NOT a polynomial-optics database lens")
    path = os.path.join(os.path.dirname(__file__), "..", "pota_amd", "lenses", name + ".json")
    with open(path, "w") as f:
        json.dump(table, f, indent=1)
    print(f"  efl {efl:.3f} mm  f/{fstop:.2f}  length {lens.z_sensor:.3f}  -> {os.path.normpath(path)}")


if __name__ == "__main__":
    names = sys.argv[1:] or list(LENSES)
    for nm in names:
        build(nm, LENSES[nm])
