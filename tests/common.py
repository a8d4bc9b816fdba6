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


def run_oracle(lib, p, table, visits, n_aovs=1, bokeh=None, keep_log=True, kinds=None, threads=None):
    """The oracle over a stream, into one frame.  threads: 1 = the single-threaded walk, visit after visit (src/lentil_filter.cpp's
    own order); None = over this machine's CPUs where the stream is a uniform one of some size and the frame has no lentil_debug
    AOV (orc_redistribute_threads: counters, draw log and every pixel no draw reaches are the single-threaded walk's bit for
    bit, the draws are added in visit order behind the pixels' own visits -- inside the 1e-5 bar every comparison of draws'
    sums is made at), else single-threaded."""
    lens = lib.orc_lens_create(C.byref(table)) if table is not None else None
    fr = oracle_lib.Frame(lib, p, n_aovs=n_aovs, kinds=kinds, keep_log=keep_log)
    n_threads = threads if threads is not None else (min(os.cpu_count() or 1, 32) if int(visits.n) >= 20000 else 1)
    rv = int(visits.pixels_per_row) * int(visits.visits_per_pixel)
    if not (n_threads > 1 and rv > 0 and fr.run_threads(lens, bokeh, visits, n_threads, rv)):
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


class ThreadedOracle:
    """The oracle over a whole (large) stream, threaded over contiguous visit ranges that end on pixel-row boundaries.

    One frame, shared (oracle/lentil_oracle.cpp: orc_redistribute_threads): a visit that stays in its pixel is added by the one
    thread that owns the pixel's row, in iterator order -- bit for bit what a single thread leaves there; the accepted draws
    are kept as records by the threads and added afterwards in visit order (inside the 1e-5 bar, and the fp64 shadows add
    up exactly).  (The reference's own threads share their buffers and race)  Streams that form does not take
    (lentil_debug, cryptomatte, thin-lens abb_chromatic > 0) fall back to one private frame per thread, merged in thread order
    -- 60 B per pixel, AOV and thread, which is what used to bound the thread count of every 4K test."""

    def __init__(self, lib, p, table, visits, n_threads, n_aovs=1, kinds=None, bokeh=None, row_visits=None, probe=None):
        import threading
        self.lib = lib
        lens = lib.orc_lens_create(C.byref(table)) if table is not None else None
        n = int(visits.n)
        rv = int(row_visits or (visits.pixels_per_row * visits.visits_per_pixel))
        rows = (n + rv - 1) // rv
        n_threads = max(1, min(n_threads, rows))
        shared = oracle_lib.Frame(lib, p, n_aovs=n_aovs, kinds=kinds, keep_log=True)
        if probe is not None:          # (fn address, user address[, camera_to_world]): the occlusion probe, oracle_lib.Frame.set_probe
            shared.set_probe(*probe)
        try:
            taken = rv > 0 and shared.run_threads(lens, bokeh, visits, max(n_threads, min(os.cpu_count() or 1, rows, 64)), rv)
        except Exception:
            shared.close()
            if lens:
                lib.orc_lens_destroy(lens)
            raise
        if taken:
            if lens:
                lib.orc_lens_destroy(lens)
            self.frames = [shared]
            self._log = shared.log()
            return
        shared.close()
        bounds = [min(n, int(round(i * rows / n_threads)) * rv) for i in range(n_threads + 1)]
        self.frames = [oracle_lib.Frame(lib, p, n_aovs=n_aovs, kinds=kinds, keep_log=True) for _ in range(n_threads)]
        if probe is not None:
            for f in self.frames:
                f.set_probe(*probe)
        errs = []

        def work(i):
            try:
                self.frames[i].run(lens, bokeh, visits, bounds[i], bounds[i + 1])
            except Exception as e:      # noqa: BLE001
                errs.append(e)

        ts = [threading.Thread(target=work, args=(i,)) for i in range(n_threads)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if lens:
            lib.orc_lens_destroy(lens)
        if errs:
            raise errs[0]
        self._log = np.concatenate([f.log() for f in self.frames], axis=0)
        for f in self.frames[1:]:
            lib.orc_frame_merge(self.frames[0].h, f.h)
            f.close()
        self.frames = self.frames[:1]

    def __getattr__(self, name):         # buffer / weight / buffer64 / weight64 / resolve / counters of the merged frame
        if name == "frames":
            raise AttributeError(name)
        return getattr(self.frames[0], name)

    def log(self):
        return self._log

    def close(self):
        for f in self.frames:
            f.close()


def host_memory_gb():
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable:"):
                    return int(line.split()[1]) / (1 << 20)
    except OSError:
        pass
    return 16.0
