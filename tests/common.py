"""Shared scaffolding of the parity tests: build params / lens / visit streams, run the oracle."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from pota_amd import _abi, camera, capi, lens_io, workload  # noqa: E402

import oracle_lib  # noqa: E402


def po_setup(width, height, lens="double_gauss_50mm", aa=3, filter_width=1.0, samples_override=0,
             focus_dist=150.0, **kw):
    p = camera.default_params()
    camera.setup_filter(p, width, height, filter_width=filter_width, aa_samples=aa)
    p, model = camera.setup_po(p, lens, focus_dist=focus_dist)
    p.samples_override = samples_override
    for k, v in kw.items():
        setattr(p, k, v)
    table, keep = lens_io.make_lens_table(model.spec)
    return p, model, table, keep


def tl_setup(width, height, aa=3, filter_width=1.0, samples_override=0, **kw):
    p = camera.default_params()
    camera.setup_filter(p, width, height, filter_width=filter_width, aa_samples=aa)
    camera.setup_thinlens(p)
    p.samples_override = samples_override
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def tan_half_fov(p):
    return float(p.sensor_width) * 0.5 / float(p.focal_length)


def make_stream(p, width, height, M, f_hi, n_extra=0, seed=0x5EED, v_end=None, **kw):
    n = width * height * M if v_end is None else v_end
    cols = workload.generate(np, 0, n, width, height, M, seed=seed, f_hi=f_hi,
                             focus_dist=float(p.focus_distance) / (10.0 if p.cameraType == 1 else 1.0),
                             tan_half_fov=tan_half_fov(p), n_extra=n_extra, **kw)
    visits, keep = capi.make_visits(cols, visits_per_pixel=M, pixels_per_row=width,
                                    pixel_row_stride=kw.get("row_stride", 1), pixel_y0=kw.get("row_offset", 0))
    return visits, cols


def run_oracle(lib, p, table, visits, n_aovs=1, bokeh=None, keep_log=True):
    lens = lib.orc_lens_create(C.byref(table)) if table is not None else None
    fr = oracle_lib.Frame(lib, p, n_aovs=n_aovs, keep_log=keep_log)
    fr.run(lens, bokeh, visits)
    if lens:
        lib.orc_lens_destroy(lens)
    return fr


def sort_log(rec):
    if rec.shape[0] == 0:
        return rec
    order = np.lexsort((rec[:, 1], rec[:, 0]))
    return rec[order]


def rel_err(a, b, floor=1e-30):
    """max |a-b| / max(|b|, floor-scaled) over elements (b = reference)."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    scale = np.maximum(np.abs(b), floor)
    return float(np.max(np.abs(a - b) / scale)) if a.size else 0.0
