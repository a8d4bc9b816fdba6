"""Bokeh-image importance tables (host setup), via liblentil_host.so."""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_PKG, "liblentil_host.so")
_host = None


def load_host_library():
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise FileNotFoundError("%s is missing: run __graft_entry__.build()" % HOST_LIB_PATH)
        _host = C.CDLL(HOST_LIB_PATH)
        _host.lentil_host_bokeh_probability.restype = C.c_int
        _host.lentil_host_bokeh_probability.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return _host


def build_tables(texels):
    """texels: float32 [y, x, nchannels] as AiTextureLoad would deliver -> dict of the four tables
    (imageData::bokehProbability, src/imagebokeh.h:143-338)."""
    t = np.ascontiguousarray(texels, np.float32)
    y, x, nch = t.shape
    out = dict(x=x, y=y, cdfRow=np.empty(y, np.float32), rowIndices=np.empty(y, np.int32),
               cdfColumn=np.empty(x * y, np.float32), columnIndices=np.empty(x * y, np.int32))
    rc = load_host_library().lentil_host_bokeh_probability(
        t.ctypes.data, x, y, nch, out["cdfRow"].ctypes.data, out["rowIndices"].ctypes.data,
        out["cdfColumn"].ctypes.data, out["columnIndices"].ctypes.data)
    if rc:
        raise ValueError("invalid bokeh image (must be square with >= 3 channels)")
    return out
