"""ctypes binding of liblentil_hip.so (the C-ABI declared in include/lentil_hip.h).

The product path: no CPU fallback.  If the HIP library is missing or no GPU is present the
calls fail loudly.
"""
import ctypes as C
import os

import numpy as np

from . import _abi

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LENTIL_HIP_LIB") or os.path.join(_PKG, "liblentil_hip.so")   # override: A/B runs of two builds

EXPORTS = [
    "lentil_hip_abi_version", "lentil_hip_create", "lentil_hip_destroy", "lentil_hip_last_error", "lentil_hip_last_redo_note",
    "lentil_hip_set_params", "lentil_hip_set_lens", "lentil_hip_set_bokeh", "lentil_hip_alloc_frame", "lentil_hip_set_camera_motion", "lentil_hip_set_camera_shutter",
    "lentil_hip_upload_visits", "lentil_hip_bind_visits", "lentil_hip_clear_frame",
    "lentil_hip_redistribute", "lentil_hip_resolve", "lentil_hip_sync", "lentil_hip_download_aov",
    "lentil_hip_download_accum", "lentil_hip_download_records", "lentil_hip_accum_buffer", "lentil_hip_stream",
    "lentil_hip_set_closest_exchange", "lentil_hip_zkey_buffer", "lentil_hip_closest_gather",
    "lentil_hip_touched_rows", "lentil_hip_merge_rows", "lentil_hip_resolve_rows",
    "lentil_hip_pack_rows", "lentil_hip_merge_packed_rows", "lentil_hip_compact_rows", "lentil_hip_merge_sparse",
    "lentil_hip_get_counters", "lentil_hip_last_timing", "lentil_hip_last_launches", "lentil_hip_set_draw_log",
    "lentil_hip_batch_model_stats", "lentil_hip_process_stats", "lentil_hip_process_stall_notes", "lentil_hip_set_async", "lentil_hip_pass_totals", "lentil_hip_set_occlusion_probe", "lentil_hip_probe_stats", "lentil_hip_debug_batch_estimate", "lentil_hip_box_probe",
    "lentil_hip_lens_jit_status", "lentil_hip_lens_jit_wait", "lentil_hip_debug_lens_jit_source", "lentil_hip_debug_lens_jit_compile",
    "lentil_hip_download_draw_log", "lentil_hip_test_lt_sample_aperture",
    "lentil_hip_test_trace_bw_po", "lentil_hip_test_aperture_sample", "lentil_hip_debug_scan_bands",
    "lentil_hip_lens_is_compiled", "lentil_hip_set_lens_mode",
    "lentil_hip_focus_search", "lentil_hip_test_y0_intersection",
    "lentil_hip_set_xor128_state", "lentil_hip_get_xor128_state",
    "lentil_hip_host_alloc", "lentil_hip_host_free", "lentil_hip_visits_begin", "lentil_hip_visits_append",
    "lentil_hip_visits_wait", "lentil_hip_visits_end",
    "lentil_hip_comm_unique_id", "lentil_hip_comm_init", "lentil_hip_comm_destroy", "lentil_hip_allreduce",
    "lentil_hip_exchange_bands", "lentil_hip_exchange_stats", "lentil_hip_exchange_counts", "lentil_hip_streams_concurrent",
    "lentil_hip_alloc_crypto", "lentil_hip_upload_crypto", "lentil_hip_bind_crypto", "lentil_hip_download_crypto",
    "lentil_hip_download_crypto_table", "lentil_hip_visits_begin_crypto", "lentil_hip_visits_append_crypto",
]

_lib = None


def process_stats():
    """(streamed passes begun, passes that hit the stuck time-out, ... of which injected, passes wiped and run again) -- all contexts of this process"""
    lib = load_library()
    n = (C.c_uint64 * 4)()
    rc = lib.lentil_hip_process_stats(n)
    if rc:
        raise RuntimeError("lentil_hip_process_stats: %d" % rc)
    return tuple(int(x) for x in n)


def process_stall_notes():
    """the redo notes of this process's passes that hit the stuck time-out unasked (lentil_hip_process_stall_notes)"""
    lib = load_library()
    buf = C.create_string_buffer(1 << 14)
    lib.lentil_hip_process_stall_notes(buf, 1 << 14)
    return buf.value.decode(errors="replace")


class LentilError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("lentil_hip error %d: %s" % (code, msg))
        self.code = code


def load_library():
    """dlopen liblentil_hip.so; raises if it has not been built (python __graft_entry__.py)."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 (same SONAME as
    # /opt/rocm's).  If torch is going to be used in this process (device memory, RCCL) it must be
    # loaded first so that this library binds to the runtime torch uses; loading /opt/rocm's runtime
    # first leaves torch without a visible GPU.
    if os.environ.get("LENTIL_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, u32, u64, i = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
    sig = {
        "lentil_hip_abi_version": (i, []),
        "lentil_hip_create": (i, [i, C.POINTER(vp)]),
        "lentil_hip_destroy": (i, [vp]),
        "lentil_hip_last_error": (C.c_char_p, [vp]),
        "lentil_hip_last_redo_note": (C.c_char_p, [vp]),
        "lentil_hip_set_params": (i, [vp, C.POINTER(_abi.Params)]),
        "lentil_hip_set_lens": (i, [vp, C.POINTER(_abi.LensTable)]),
        "lentil_hip_set_bokeh": (i, [vp, C.POINTER(_abi.BokehTable)]),
        "lentil_hip_alloc_frame": (i, [vp, u32, vp]),
        "lentil_hip_set_camera_motion": (i, [vp, u32, vp]),
        "lentil_hip_set_camera_shutter": (i, [vp, C.c_float, C.c_float]),
        "lentil_hip_upload_visits": (i, [vp, C.POINTER(_abi.Visits)]),
        "lentil_hip_bind_visits": (i, [vp, C.POINTER(_abi.Visits)]),
        "lentil_hip_clear_frame": (i, [vp]),
        "lentil_hip_redistribute": (i, [vp]),
        "lentil_hip_resolve": (i, [vp]),
        "lentil_hip_sync": (i, [vp]),
        "lentil_hip_download_aov": (i, [vp, u32, vp]),
        "lentil_hip_download_accum": (i, [vp, u32, vp, vp]),
        "lentil_hip_download_records": (i, [vp, vp, u64, C.POINTER(u32)]),
        "lentil_hip_accum_buffer": (i, [vp, C.POINTER(vp), C.POINTER(u64)]),
        "lentil_hip_stream": (i, [vp, C.POINTER(vp)]),
        "lentil_hip_set_closest_exchange": (i, [vp, i, u32]),
        "lentil_hip_zkey_buffer": (i, [vp, C.POINTER(vp), C.POINTER(u64)]),
        "lentil_hip_closest_gather": (i, [vp]),
        "lentil_hip_touched_rows": (i, [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
        "lentil_hip_merge_rows": (i, [vp, u32, u32, vp, vp]),
        "lentil_hip_pack_rows": (i, [vp, u32, u32, vp]),
        "lentil_hip_merge_packed_rows": (i, [vp, u32, u32, vp, vp]),
        "lentil_hip_compact_rows": (i, [vp, u32, u32, vp, vp, vp, u32, C.POINTER(u32)]),
        "lentil_hip_merge_sparse": (i, [vp, u32, u32, u32, vp, vp, vp]),
        "lentil_hip_resolve_rows": (i, [vp, u32, u32]),
        "lentil_hip_get_counters": (i, [vp, C.POINTER(_abi.Counters)]),
        "lentil_hip_last_timing": (i, [vp, C.POINTER(C.c_float)]),
        "lentil_hip_last_launches": (i, [vp, C.POINTER(C.c_uint32)]),
        "lentil_hip_batch_model_stats": (i, [vp, C.POINTER(C.c_uint64)]),
        "lentil_hip_process_stats": (i, [C.POINTER(C.c_uint64)]),
        "lentil_hip_process_stall_notes": (i, [C.c_char_p, C.c_uint64]),
        "lentil_hip_set_async": (i, [vp, i]),
        "lentil_hip_set_occlusion_probe": (i, [vp, vp, vp, vp]),
        "lentil_hip_probe_stats": (i, [vp, C.POINTER(C.c_uint64)]),
        "lentil_hip_pass_totals": (i, [vp, C.POINTER(_abi.PassTotals), i]),
        "lentil_hip_box_probe": (i, [vp, C.POINTER(C.c_double)]),
        "lentil_hip_lens_jit_status": (i, [vp, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
        "lentil_hip_lens_jit_wait": (i, [vp, C.c_double]),
        "lentil_hip_debug_lens_jit_compile": (i, [vp, i, C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_char_p, C.c_uint64,
                                                  C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
        "lentil_hip_debug_lens_jit_source": (i, [vp, C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64)]),
        "lentil_hip_debug_batch_estimate": (i, [vp, C.c_uint64, vp, C.c_uint32, vp]),
        "lentil_hip_alloc_crypto": (i, [vp, u32, u32]),
        "lentil_hip_upload_crypto": (i, [vp, C.POINTER(_abi.CryptoVisits)]),
        "lentil_hip_bind_crypto": (i, [vp, C.POINTER(_abi.CryptoVisits)]),
        "lentil_hip_download_crypto": (i, [vp, u32, u32, vp, vp]),
        "lentil_hip_visits_begin_crypto": (i, [vp, u32]),
        "lentil_hip_visits_append_crypto": (i, [vp, C.POINTER(_abi.Visits), C.POINTER(_abi.CryptoVisits), C.POINTER(u64)]),
        "lentil_hip_download_crypto_table": (i, [vp, u32, C.POINTER(u32), vp, vp, vp]),
        "lentil_hip_set_draw_log": (i, [vp, u64]),
        "lentil_hip_download_draw_log": (i, [vp, vp, u64, C.POINTER(u64)]),
        "lentil_hip_test_lt_sample_aperture": (i, [vp, u64, vp, vp, C.c_double, vp, vp, vp]),
        "lentil_hip_test_trace_bw_po": (i, [vp, u64, vp, vp, vp, vp, vp, vp]),
        "lentil_hip_test_aperture_sample": (i, [vp, u64, vp, vp, vp]),
        "lentil_hip_lens_is_compiled": (i, [vp]),
        "lentil_hip_set_lens_mode": (i, [vp, i]),
        "lentil_hip_focus_search": (i, [vp, C.c_double, C.c_double, C.POINTER(C.c_double)]),
        "lentil_hip_set_xor128_state": (i, [vp, C.POINTER(C.c_uint32)]),
        "lentil_hip_get_xor128_state": (i, [vp, C.POINTER(C.c_uint32)]),
        "lentil_hip_test_y0_intersection": (i, [vp, u64, vp, C.c_double, vp, vp, vp]),
        "lentil_hip_host_alloc": (i, [C.POINTER(vp), u64]),
        "lentil_hip_host_free": (i, [vp]),
        "lentil_hip_visits_begin": (i, [vp, C.POINTER(_abi.Visits), u64]),
        "lentil_hip_visits_append": (i, [vp, C.POINTER(_abi.Visits), C.POINTER(u64)]),
        "lentil_hip_visits_wait": (i, [vp, u64]),
        "lentil_hip_visits_end": (i, [vp, C.POINTER(u64)]),
        "lentil_hip_comm_unique_id": (i, [vp]),
        "lentil_hip_comm_init": (i, [vp, vp, i, i]),
        "lentil_hip_comm_destroy": (i, [vp]),
        "lentil_hip_allreduce": (i, [vp]),
        "lentil_hip_exchange_bands": (i, [vp, vp, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
        "lentil_hip_exchange_stats": (i, [vp, C.POINTER(u64), C.POINTER(u64)]),
        "lentil_hip_exchange_counts": (i, [vp, C.POINTER(u64), C.POINTER(u64)]),
        "lentil_hip_streams_concurrent": (i, [vp, C.POINTER(C.c_int)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.lentil_hip_abi_version() != 1:
        raise RuntimeError("liblentil_hip.so ABI version mismatch")
    _lib = lib
    return lib


def host_alloc(nbytes):
    """Page-locked host memory from the library (lentil_hip_host_alloc) as a ctypes char array; free with host_free."""
    p = C.c_void_p()
    rc = load_library().lentil_hip_host_alloc(C.byref(p), nbytes)
    if rc:
        raise LentilError(rc, "lentil_hip_host_alloc(%d)" % nbytes)
    return p.value


def host_free(ptr):
    load_library().lentil_hip_host_free(C.c_void_p(ptr))


def make_crypto_visits(hashes, weights, ptr=lambda a: a.ctypes.data):
    """lentil_crypto_visits from per-AOV lists of [n, entries] fp32 arrays.  Returns (CryptoVisits, keepalive)."""
    c = _abi.CryptoVisits()
    c.n_crypto = len(hashes)
    c.n, c.entries = hashes[0].shape
    keep = []
    for a, (h, w) in enumerate(zip(hashes, weights)):
        assert h.shape == w.shape == (c.n, c.entries)
        c.hash[a] = ptr(h)
        c.weight[a] = ptr(w)
        keep += [h, w]
    return c, keep


def make_visits(cols, visits_per_pixel=0, pixels_per_row=0, pixel_x0=0, pixel_y0=0, pixel_row_stride=1,
                ptr=lambda a: a.ctypes.data):
    """Fill a lentil_visits from a dict of columns (numpy arrays, or anything `ptr` understands).

    cols: rgba, pos_z, raydir_time, volume_ignore, transmission [n,4] fp32; optional extra (list),
    pixel (uint32 [n]), inv_density (fp32 [n]).  Returns (Visits, keepalive)."""
    v = _abi.Visits()
    n = int(cols["rgba"].shape[0])
    v.n = n
    v.visits_per_pixel = visits_per_pixel
    v.pixels_per_row = pixels_per_row
    v.pixel_x0, v.pixel_y0 = pixel_x0, pixel_y0
    v.pixel_row_stride = pixel_row_stride
    extra = list(cols.get("extra", []))
    v.n_extra = len(extra)
    for name in ("rgba", "pos_z", "raydir_time", "volume_ignore", "transmission"):
        setattr(v, name, ptr(cols[name]) if n else None)
    for k, e in enumerate(extra):
        v.extra[k] = ptr(e) if (n and e is not None) else None      # None: an AOV without a column (lentil_debug)
    if cols.get("pixel") is not None:
        v.pixel = ptr(cols["pixel"])
    if cols.get("inv_density") is not None:
        v.inv_density = ptr(cols["inv_density"])
    return v, cols


def lens_jit_compile(table, compile=True):
    """(rc, emitted source, compiler log, seconds, code bytes) -- no context, no GPU (lentil_hip_debug_lens_jit_compile)"""
    lib = load_library()
    n = C.c_uint64(0)
    lib.lentil_hip_debug_lens_jit_compile(C.byref(table), 0, None, 0, C.byref(n), None, 0, None, None)
    src = C.create_string_buffer(int(n.value) + 1)
    log = C.create_string_buffer(1 << 16)
    sec, nb = C.c_double(0.0), C.c_uint64(0)
    rc = lib.lentil_hip_debug_lens_jit_compile(C.byref(table), 1 if compile else 0, src, int(n.value) + 1, None, log, 1 << 16, C.byref(sec), C.byref(nb))
    return int(rc), src.value.decode(), log.value.decode(errors="replace"), float(sec.value), int(nb.value)


class Context:
    """One lentil_hip_ctx: one GPU, one stream (mirrors the ownership of the reference's Camera)."""

    def __init__(self, device=0):
        self.lib = load_library()
        self.h = C.c_void_p()
        rc = self.lib.lentil_hip_create(device, C.byref(self.h))
        if rc:
            raise LentilError(rc, (self.lib.lentil_hip_last_error(None) or b"").decode())
        self.params = None
        self.n_aovs = 0
        self._keep = {}

    def _chk(self, rc):
        if rc:
            raise LentilError(rc, (self.lib.lentil_hip_last_error(self.h) or b"").decode())

    def close(self):
        if self.h:
            self.lib.lentil_hip_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- setup
    def set_params(self, params):
        self._chk(self.lib.lentil_hip_set_params(self.h, C.byref(params)))
        self.params = params

    def set_lens(self, table):
        self._chk(self.lib.lentil_hip_set_lens(self.h, C.byref(table)))

    def lens_is_compiled(self):
        return bool(self.lib.lentil_hip_lens_is_compiled(self.h))

    def set_lens_mode(self, mode):
        """0 = compiled-in kernel when the table is a shipped lens, 1 = always the table interpreter."""
        self._chk(self.lib.lentil_hip_set_lens_mode(self.h, mode))

    def set_bokeh(self, tables):
        """tables: dict with x, y, cdfRow, rowIndices, cdfColumn, columnIndices (numpy) or None."""
        if tables is None:
            self._chk(self.lib.lentil_hip_set_bokeh(self.h, None))
            return
        b = _abi.BokehTable()
        b.x, b.y = int(tables["x"]), int(tables["y"])
        arrs = {k: np.ascontiguousarray(tables[k]) for k in ("cdfRow", "rowIndices", "cdfColumn", "columnIndices")}
        assert arrs["cdfRow"].dtype == np.float32 and arrs["rowIndices"].dtype == np.int32
        assert arrs["cdfColumn"].dtype == np.float32 and arrs["columnIndices"].dtype == np.int32
        for k, a in arrs.items():
            setattr(b, k, a.ctypes.data)
        self._chk(self.lib.lentil_hip_set_bokeh(self.h, C.byref(b)))

    def alloc_frame(self, n_aovs=1, kinds=None):
        k = (C.c_uint8 * n_aovs)(*(kinds or [0] * n_aovs))
        self._chk(self.lib.lentil_hip_alloc_frame(self.h, n_aovs, C.cast(k, C.c_void_p)))
        self.n_aovs = n_aovs

    # --- visits
    def set_camera_shutter(self, start, end):
        """the absolute times of the first and the last camera matrix key (Arnold's shutter_start / shutter_end); 0 ... 1 until set"""
        self._chk(self.lib.lentil_hip_set_camera_shutter(self.h, float(start), float(end)))

    def set_camera_motion(self, keys):
        """[n, 4, 4] world-to-camera matrices at equidistant times over the shutter (None / one key: the static matrix)"""
        if keys is None:
            self._chk(self.lib.lentil_hip_set_camera_motion(self.h, 0, None))
            return
        k = np.ascontiguousarray(keys, np.float32)
        self._chk(self.lib.lentil_hip_set_camera_motion(self.h, k.shape[0], k.ctypes.data))

    def upload_visits(self, visits):
        self._chk(self.lib.lentil_hip_upload_visits(self.h, C.byref(visits)))

    def bind_visits(self, visits, keepalive=None):
        self._chk(self.lib.lentil_hip_bind_visits(self.h, C.byref(visits)))
        self._keep["visits"] = keepalive

    # --- hot path
    def clear_frame(self):
        self._chk(self.lib.lentil_hip_clear_frame(self.h))

    def redistribute(self):
        self._chk(self.lib.lentil_hip_redistribute(self.h))

    def resolve(self):
        self._chk(self.lib.lentil_hip_resolve(self.h))

    def sync(self):
        self._chk(self.lib.lentil_hip_sync(self.h))

    # --- results
    @property
    def n_pixels(self):
        return int(self.params.xres) * int(self.params.yres)

    def download_aov(self, aov=0):
        out = np.empty((self.n_pixels, 4), np.float32)
        self._chk(self.lib.lentil_hip_download_aov(self.h, aov, out.ctypes.data))
        return out

    def download_accum(self, aov=0):
        buf = np.empty((self.n_pixels, 4), np.float32)
        w = np.empty((self.n_pixels,), np.float32)
        self._chk(self.lib.lentil_hip_download_accum(self.h, aov, buf.ctypes.data, w.ctypes.data))
        return buf, w

    def download_records(self):
        """[n_pixels, stride] fp32: every AOV's accumulators (floats 4a .. 4a+3) and the weight (float 4 * n_aovs) in one copy"""
        st = C.c_uint32(0)
        self._chk(self.lib.lentil_hip_download_records(self.h, None, 0, C.byref(st)))
        rec = np.empty((self.n_pixels, int(st.value)), np.float32)
        self._chk(self.lib.lentil_hip_download_records(self.h, rec.ctypes.data, rec.size, C.byref(st)))
        return rec

    def accum_buffer(self):
        p, n = C.c_void_p(), C.c_uint64()
        self._chk(self.lib.lentil_hip_accum_buffer(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def set_closest_exchange(self, deferred, visit_id_base=0):
        self._chk(self.lib.lentil_hip_set_closest_exchange(self.h, 1 if deferred else 0, visit_id_base))

    def zkey_buffer(self):
        p, n = C.c_void_p(), C.c_uint64()
        self._chk(self.lib.lentil_hip_zkey_buffer(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def closest_gather(self):
        self._chk(self.lib.lentil_hip_closest_gather(self.h))

    def set_xor128_state(self, state):
        self._chk(self.lib.lentil_hip_set_xor128_state(self.h, (C.c_uint32 * 4)(*[int(x) for x in state])))

    def get_xor128_state(self):
        st = (C.c_uint32 * 4)()
        self._chk(self.lib.lentil_hip_get_xor128_state(self.h, st))
        return list(st)

    def focus_search(self, focal_distance, lam):
        best = C.c_double()
        self._chk(self.lib.lentil_hip_focus_search(self.h, focal_distance, lam, C.byref(best)))
        return best.value

    def test_y0_intersection(self, sensor_shift, lam):
        import numpy as np
        sh = np.ascontiguousarray(sensor_shift, np.float64)
        n = sh.shape[0]
        dist, sensor, out = np.empty(n), np.empty((n, 5)), np.empty((n, 5))
        self._chk(self.lib.lentil_hip_test_y0_intersection(self.h, n, sh.ctypes.data, lam, dist.ctypes.data,
                                                           sensor.ctypes.data, out.ctypes.data))
        return dist, sensor, out

    # --- the visit stream handed over piece by piece (include/lentil_hip.h "piece by piece")
    def visits_begin(self, layout, capacity_hint=0):
        self._chk(self.lib.lentil_hip_visits_begin(self.h, C.byref(layout), capacity_hint))

    def visits_append(self, part):
        t = C.c_uint64()
        self._chk(self.lib.lentil_hip_visits_append(self.h, C.byref(part), C.byref(t)))
        return t.value

    def visits_begin_crypto(self, entries):
        self._chk(self.lib.lentil_hip_visits_begin_crypto(self.h, entries))

    def visits_append_crypto(self, part, caches):
        t = C.c_uint64()
        self._chk(self.lib.lentil_hip_visits_append_crypto(self.h, C.byref(part), C.byref(caches), C.byref(t)))
        return t.value

    def visits_wait(self, ticket):
        self._chk(self.lib.lentil_hip_visits_wait(self.h, ticket))

    def visits_end(self):
        n = C.c_uint64()
        self._chk(self.lib.lentil_hip_visits_end(self.h, C.byref(n)))
        return n.value

    # --- the native exchange (RCCL inside the library; include/lentil_hip.h "multi-GPU, the exchange itself")
    @staticmethod
    def comm_unique_id():
        buf = (C.c_uint8 * 128)()
        lib = load_library()
        rc = lib.lentil_hip_comm_unique_id(buf)
        if rc != 0:
            raise RuntimeError("lentil_hip_comm_unique_id: %s" % (lib.lentil_hip_last_error(None) or b"").decode())
        return bytes(buf)

    def comm_init(self, uid, rank, world):
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        self._chk(self.lib.lentil_hip_comm_init(self.h, buf, rank, world))

    def comm_destroy(self):
        self._chk(self.lib.lentil_hip_comm_destroy(self.h))

    def allreduce(self):
        self._chk(self.lib.lentil_hip_allreduce(self.h))

    def exchange_bands(self, visit_rows, bounds=None, sparse=True):
        lo, hi = C.c_int32(), C.c_int32()
        b = None
        if bounds is not None:
            b = (C.c_int32 * len(bounds))(*[int(x) for x in bounds])
        self._chk(self.lib.lentil_hip_exchange_bands(self.h, b, int(visit_rows), 1 if sparse else 0,
                                                     C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def streams_concurrent(self):
        """0: the pass's four streams share hardware queues, 1: they run side by side, 2: and a spare one beside them"""
        v = C.c_int(0)
        self._chk(self.lib.lentil_hip_streams_concurrent(self.h, C.byref(v)))
        return int(v.value)

    def exchange_stats(self):
        """(bytes sent, bytes received) by this rank in its last exchange_bands / allreduce"""
        a, b = C.c_uint64(), C.c_uint64()
        self._chk(self.lib.lentil_hip_exchange_stats(self.h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def exchange_counts(self):
        """(exchange_bands calls that ran in the fixed-capacity form, directed pairs whose entries overflowed their message)"""
        a, b = C.c_uint64(), C.c_uint64()
        self._chk(self.lib.lentil_hip_exchange_counts(self.h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def touched_rows(self):
        lo, hi = C.c_int32(), C.c_int32()
        self._chk(self.lib.lentil_hip_touched_rows(self.h, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def merge_rows(self, row_begin, n_rows, acc_ptr, key_ptr=None):
        self._chk(self.lib.lentil_hip_merge_rows(self.h, row_begin, n_rows, acc_ptr, key_ptr))

    def pack_rows(self, row_begin, n_rows, dst_ptr):
        self._chk(self.lib.lentil_hip_pack_rows(self.h, row_begin, n_rows, dst_ptr))

    def merge_packed_rows(self, row_begin, n_rows, packed_ptr, key_ptr=None):
        self._chk(self.lib.lentil_hip_merge_packed_rows(self.h, row_begin, n_rows, packed_ptr, key_ptr))

    def compact_rows(self, row_begin, n_rows, idx_ptr, vals_ptr, keys_ptr, capacity):
        n = C.c_uint32()
        self._chk(self.lib.lentil_hip_compact_rows(self.h, row_begin, n_rows, idx_ptr, vals_ptr, keys_ptr, capacity, C.byref(n)))
        return n.value

    def merge_sparse(self, row_begin, n_rows, n, idx_ptr, vals_ptr, keys_ptr=None):
        self._chk(self.lib.lentil_hip_merge_sparse(self.h, row_begin, n_rows, n, idx_ptr, vals_ptr, keys_ptr))

    def resolve_rows(self, row_begin, n_rows):
        self._chk(self.lib.lentil_hip_resolve_rows(self.h, row_begin, n_rows))

    def stream(self):
        s = C.c_void_p()
        self._chk(self.lib.lentil_hip_stream(self.h, C.byref(s)))
        return s.value

    def last_redo_note(self):
        """Why the last streamed pass that was redone the chunked way gave up ("" if none was)."""
        return (self.lib.lentil_hip_last_redo_note(self.h) or b"").decode()

    def counters(self):
        c = _abi.Counters()
        self._chk(self.lib.lentil_hip_get_counters(self.h, C.byref(c)))
        return c

    def set_occlusion_probe(self, fn, user=None, camera_to_world=None):
        """fn: address of a lentil_probe_fn (C function; None switches probing off); user: address handed back to it;
        camera_to_world: [4, 4] fp32 (row-vector convention) or None for the inverse of the world-to-camera matrix"""
        self._probe_keep = (np.ascontiguousarray(camera_to_world, np.float32) if camera_to_world is not None else None)
        self._chk(self.lib.lentil_hip_set_occlusion_probe(self.h, fn, user, self._probe_keep.ctypes.data if self._probe_keep is not None else None))

    def probe_stats(self):
        """(segments probed, of them occluded, callback calls) since the context was created"""
        n = (C.c_uint64 * 3)()
        self._chk(self.lib.lentil_hip_probe_stats(self.h, n))
        return tuple(int(x) for x in n)

    def set_async(self, on):
        """the asynchronous end of a pass (include/lentil_hip.h): off = every redistribute waits for its own end"""
        self._chk(self.lib.lentil_hip_set_async(self.h, 1 if on else 0))

    def pass_totals(self, reset=False):
        """sums over the passes looked at since the last reset (observes the context: waits for what is in flight)"""
        t = _abi.PassTotals()
        self._chk(self.lib.lentil_hip_pass_totals(self.h, C.byref(t), 1 if reset else 0))
        return t

    def last_timing(self):
        ms = (C.c_float * 3)()
        self._chk(self.lib.lentil_hip_last_timing(self.h, ms))
        return float(ms[0]), float(ms[1]), float(ms[2])

    def last_launches(self):
        n = (C.c_uint32 * 2)()
        self._chk(self.lib.lentil_hip_last_launches(self.h, n))
        return int(n[0]), int(n[1])

    def lens_jit_status(self):
        """(state, compile seconds): 0 nothing to compile, 1 compiling, 2 specialised kernel in use, -1 failed"""
        st, sec = C.c_int(0), C.c_double(0.0)
        self._chk(self.lib.lentil_hip_lens_jit_status(self.h, C.byref(st), C.byref(sec)))
        return int(st.value), float(sec.value)

    def lens_jit_wait(self, timeout_seconds=0.0):
        self._chk(self.lib.lentil_hip_lens_jit_wait(self.h, float(timeout_seconds)))

    def lens_jit_source(self):
        n = C.c_uint64(0)
        self._chk(self.lib.lentil_hip_debug_lens_jit_source(self.h, None, 0, C.byref(n)))
        buf = C.create_string_buffer(int(n.value) + 1)
        self._chk(self.lib.lentil_hip_debug_lens_jit_source(self.h, buf, int(n.value) + 1, None))
        return buf.value.decode()

    def box_probe(self):
        """what the GPU delivers right now: fp64 chain rate, shader clock, copy / read bandwidth, hardware queues, CUs"""
        p = (C.c_double * 6)()
        self._chk(self.lib.lentil_hip_box_probe(self.h, p))
        return {"fp64_mul_add_tflops": round(p[0], 2), "shader_clock_mhz_under_fp64": round(p[1], 1), "copy_gbs": round(p[2], 1),
                "read_gbs": round(p[3], 1), "gpu_max_hw_queues": int(p[4]), "compute_units": int(p[5])}

    def batch_model_stats(self):
        """(calibrations, lean passes, lean passes that needed a second round after all, margin sixteenths)"""
        n = (C.c_uint64 * 4)()
        self._chk(self.lib.lentil_hip_batch_model_stats(self.h, n))
        return tuple(int(x) for x in n)

    def debug_batch_estimate(self, cs_xyz, samples):
        """[n, 4]: share well inside the frame, share inside, share vignetted, first batch -- for camera-space points [n, 3]"""
        import numpy as np
        cs = np.ascontiguousarray(cs_xyz, np.float32)
        out = np.zeros((cs.shape[0], 4), np.float32)
        self._chk(self.lib.lentil_hip_debug_batch_estimate(self.h, cs.shape[0], cs.ctypes.data, int(samples), out.ctypes.data))
        return out

    # --- cryptomatte AOVs
    def alloc_crypto(self, n_crypto, slots_per_pixel=0):
        self._chk(self.lib.lentil_hip_alloc_crypto(self.h, n_crypto, slots_per_pixel))

    def upload_crypto(self, crypto_visits):
        self._chk(self.lib.lentil_hip_upload_crypto(self.h, C.byref(crypto_visits)))

    def bind_crypto(self, crypto_visits, keepalive=None):
        self._chk(self.lib.lentil_hip_bind_crypto(self.h, C.byref(crypto_visits)))
        self._keep["crypto"] = keepalive

    def download_crypto(self, crypto, rank):
        """(np x 4 RGBA, np bool: the pixel's map has more than `rank` entries)"""
        out = np.empty((self.n_pixels, 4), np.float32)
        has = np.empty(self.n_pixels, np.uint8)
        self._chk(self.lib.lentil_hip_download_crypto(self.h, crypto, rank, out.ctypes.data, has.ctypes.data))
        return out, has.astype(bool)

    def download_crypto_table(self, crypto):
        """(id bits [np, slots] uint32 with 0xFFFFFFFF = free, weights [np, slots], totals [np])"""
        slots = C.c_uint32()
        self._chk(self.lib.lentil_hip_download_crypto_table(self.h, crypto, C.byref(slots), None, None, None))
        ids = np.empty((self.n_pixels, slots.value), np.uint32)
        wts = np.empty((self.n_pixels, slots.value), np.float32)
        tot = np.empty(self.n_pixels, np.float32)
        self._chk(self.lib.lentil_hip_download_crypto_table(self.h, crypto, C.byref(slots), ids.ctypes.data, wts.ctypes.data,
                                                            tot.ctypes.data))
        return ids, wts, tot

    def set_draw_log(self, capacity):
        self._chk(self.lib.lentil_hip_set_draw_log(self.h, capacity))

    def draw_log(self):
        n = C.c_uint64()
        self._chk(self.lib.lentil_hip_download_draw_log(self.h, None, 0, C.byref(n)))
        rec = np.empty((n.value, 3), np.uint32)
        if n.value:
            self._chk(self.lib.lentil_hip_download_draw_log(self.h, rec.ctypes.data, n.value, C.byref(n)))
        return rec

    # --- primitive tests
    def test_lt_sample_aperture(self, scene, ap, lam):
        scene = np.ascontiguousarray(scene, np.float64)
        ap = np.ascontiguousarray(ap, np.float64)
        n = scene.shape[0]
        sensor = np.empty((n, 5)); out = np.empty((n, 5)); T = np.empty((n,))
        self._chk(self.lib.lentil_hip_test_lt_sample_aperture(
            self.h, n, scene.ctypes.data, ap.ctypes.data, lam, sensor.ctypes.data, out.ctypes.data, T.ctypes.data))
        return sensor, out, T

    def test_trace_bw_po(self, target, px, py, attempt):
        target = np.ascontiguousarray(target, np.float64)
        px = np.ascontiguousarray(px, np.int32); py = np.ascontiguousarray(py, np.int32)
        attempt = np.ascontiguousarray(attempt, np.int32)
        n = target.shape[0]
        xy = np.empty((n, 2)); ok = np.empty((n,), np.int32)
        self._chk(self.lib.lentil_hip_test_trace_bw_po(
            self.h, n, target.ctypes.data, px.ctypes.data, py.ctypes.data, attempt.ctypes.data,
            xy.ctypes.data, ok.ctypes.data))
        return xy, ok

    def test_aperture_sample(self, a, b):
        a = np.ascontiguousarray(a, np.uint32); b = np.ascontiguousarray(b, np.uint32)
        xy = np.empty((a.shape[0], 2))
        self._chk(self.lib.lentil_hip_test_aperture_sample(self.h, a.shape[0], a.ctypes.data, b.ctypes.data,
                                                           xy.ctypes.data))
        return xy
