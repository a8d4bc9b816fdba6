"""scan_dma2_kernel decides `get_coc_thinlens(z) < 0.4` (src/lentil.h:674-692, src/lentil_filter.cpp:185-190) from the
camera-space depth alone wherever the host's intervals (lentil_hip_debug_scan_bands) say it is certain.  No GPU: the
intervals against the oracle's fp32 evaluation of the function itself, over depths of every magnitude, both signs, the
neighbourhood of every interval end, zeros, infinities and NaN, for polynomial-optics and thin-lens cameras."""
import ctypes as C

import numpy as np
import pytest

import common
import oracle_lib
from pota_amd import _abi, camera, capi


def _bands(p):
    lib = capi.load_library()
    lib.lentil_hip_debug_scan_bands.restype = C.c_int
    lib.lentil_hip_debug_scan_bands.argtypes = [C.POINTER(_abi.Params), C.POINTER(C.c_float)]
    out = (C.c_float * 8)()
    assert lib.lentil_hip_debug_scan_bands(C.byref(p), out) == 0
    b = np.array(list(out), np.float32)
    return b[0:2], b[2:4], b[4:6], b[6:8]


def _classify(z, b):
    in_lo, in_hi, out_lo, out_hi = b
    with np.errstate(invalid="ignore"):
        inn = ((z >= in_lo[0]) & (z <= in_hi[0])) | ((z >= in_lo[1]) & (z <= in_hi[1]))
        out = ((z >= out_lo[0]) & (z <= out_hi[0])) | ((z >= out_lo[1]) & (z <= out_hi[1]))
    return np.where(inn, 1, np.where(out, 2, 0))


def _depths(b, rng):
    mags = np.float32(10.0) ** rng.uniform(-44, 38.5, 200000).astype(np.float32)
    z = np.concatenate([mags, -mags, rng.uniform(-5000, 5000, 200000).astype(np.float32)])
    ends = np.concatenate(b)
    ends = ends[np.isfinite(ends)]
    near = []
    for e in ends:
        x = np.float32(e)
        lo = x
        hi = x
        for _ in range(300):
            lo = np.nextafter(lo, np.float32(-np.inf)); hi = np.nextafter(hi, np.float32(np.inf))
            near += [lo, hi]
        near.append(x)
        near += list((np.float32(e) * (1.0 + rng.uniform(-3e-3, 3e-3, 20000))).astype(np.float32))
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 3.4e38, -3.4e38, 1e30, -1e30, 1.0000001e30], np.float32)
    return np.concatenate([z, np.array(near, np.float32), special])


CASES = [
    ("po", dict(focus_dist=150.0)), ("po", dict(focus_dist=35.0)), ("po", dict(focus_dist=4000.0)),
    ("tl", dict()), ("tl", dict(focus_distance=20.0)), ("tl", dict(focus_distance=1e6)), ("tl", dict(fstop=22.0)),
    ("tl", dict(fstop=0.8, focus_distance=300.0)), ("tl", dict(focal_length=0.8, focus_distance=55.0)),
]


@pytest.mark.parametrize("kind,kw", CASES)
def test_bands_agree_with_the_function(kind, kw):
    orc = oracle_lib.load()
    if kind == "po":
        p, model, table, keep = common.po_setup(256, 128, **kw)
    else:
        p = common.tl_setup(256, 128)
        for k, v in kw.items():
            if k == "fstop":
                p.fstop = v
                p.aperture_radius = (float(p.focal_length) / (2.0 * v))
            else:
                setattr(p, k, v)
    b = _bands(p)
    rng = np.random.default_rng(7)
    z = _depths(b, rng)
    cls = _classify(z, b)
    coc = np.array([orc.orc_get_coc_thinlens(C.byref(p), float(x)) for x in z[cls != 2]], np.float32)
    with np.errstate(invalid="ignore"):
        below = coc < np.float32(0.4)
    got = cls[cls != 2] == 1
    bad = np.nonzero(below != got)[0]
    assert bad.size == 0, "depth %r: function says below=%s (coc %r), bands say %s" % (
        z[cls != 2][bad[0]], below[bad[0]], coc[bad[0]], got[bad[0]])
    # the intervals are worth something: nearly every depth of a frame is decided without the function
    scene = rng.uniform(-3000, -5, 100000).astype(np.float32)
    assert np.mean(_classify(scene, b) == 2) < 0.02
