"""Lens tables: JSON (tools/fit_lens.py output) -> lentil_lens_table for the C-ABI.

Replaces the reference's compile-time splice of generated lens code
(include/auto_generated_lens_includes/*.h -> src/lentil.h:1262,1278,1308,1576) with
run-time tables.  The JSON keeps every coefficient with repr() precision (round-trips).
"""
import ctypes as C
import json
import os

from . import _abi

LENS_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lenses")
OUT_NAMES = ["out_x", "out_y", "out_dx", "out_dy", "out_t"]
AP_NAMES = ["ap_x", "ap_y", "ap_dx", "ap_dy"]
_GEOM = {"cyl-y": _abi.GEOM_CYL_Y, "cyl-x": _abi.GEOM_CYL_X}


def available_lenses():
    return sorted(f[:-5] for f in os.listdir(LENS_DIR) if f.endswith(".json"))


def load_lens_json(name_or_path):
    path = name_or_path if os.path.exists(name_or_path) else os.path.join(LENS_DIR, name_or_path + ".json")
    with open(path) as f:
        return json.load(f)


def make_lens_table(spec):
    """Returns (LensTable, keepalive) -- keepalive owns the term array."""
    if isinstance(spec, str):
        spec = load_lens_json(spec)
    polys = spec["polys"]
    n_terms = sum(len(polys[n]) for n in OUT_NAMES + AP_NAMES)
    terms = (_abi.Term * n_terms)()
    tab = _abi.LensTable()
    k = spec["constants"]
    for n in _abi.LENS_CONSTANT_NAMES:
        setattr(tab, n, float(k[n]))
    tab.lens_inner_pupil_geometry = _GEOM.get(k.get("lens_inner_pupil_geometry", ""), _abi.GEOM_SPHERICAL)
    tab.lens_outer_pupil_geometry = _GEOM.get(k.get("lens_outer_pupil_geometry", ""), _abi.GEOM_SPHERICAL)
    i = 0
    for dst, names in ((tab.out, OUT_NAMES), (tab.ap, AP_NAMES)):
        for j, n in enumerate(names):
            dst[j].first = i
            dst[j].count = len(polys[n])
            for c, e in polys[n]:
                terms[i].c = float(c)
                for v in range(5):
                    terms[i].e[v] = int(e[v])
                i += 1
    tab.n_terms = n_terms
    tab.terms = C.cast(terms, C.POINTER(_abi.Term))
    return tab, terms


def _ipow_mults(e):
    """multiplications of lens_ipow(x, e) (src/lens.h:226-233)"""
    if e <= 1:
        return 0
    if e == 2:
        return 1
    return _ipow_mults(e // 2) + (2 if e & 1 else 1)


def newton_iteration_flops(spec):
    """fp64 operations of one iteration of the backward Newton solve on this table, counted the way the code
    evaluates it: the 14 polynomials an iteration needs -- aperture x/y and their partials
    by dx/dy, outer pupil x/y/dx/dy and the partials of dx/dy by x/y -- as c * f(x) * f(y) * f(dx) * f(dy) * f(lambda)
    summed in term order (one multiplication per factor present, one addition per term after the first; every power
    lens_ipow(v, e) computed once per iteration), plus ~400 operations for the pupil transforms, the two 2x2
    inverses and the error terms.  Returns (multiplications, additions, other)."""
    if isinstance(spec, str):
        spec = load_lens_json(spec)
    polys = spec["polys"]
    needed = []
    for name in ("ap_x", "ap_y"):
        needed.append(polys[name])
        for var in (2, 3):
            needed.append([(c * e[var], [e[v] - (1 if v == var else 0) for v in range(5)]) for c, e in polys[name] if e[var] > 0])
    for name in ("out_x", "out_y", "out_dx", "out_dy"):
        needed.append(polys[name])
    for name in ("out_dx", "out_dy"):
        for var in (0, 1):
            needed.append([(c * e[var], [e[v] - (1 if v == var else 0) for v in range(5)]) for c, e in polys[name] if e[var] > 0])
    mul = add = 0
    powers = set()
    for poly in needed:
        for c, e in poly:
            mul += sum(1 for v in range(5) if e[v] > 0)
            for v in range(4):          # the powers of lambda are constants of the solve
                if e[v] > 1:
                    powers.add((v, e[v]))
        add += max(0, len(poly) - 1)
    mul += sum(_ipow_mults(e) for _, e in powers)
    return mul, add, 400
