"""ctypes binding of liblentil_host.so (include/lentil_host.h): CPU-side camera setup and forward
camera rays of the plugin mirror."""
import ctypes as C

import numpy as np

from . import _abi, bokeh, lens_io


class CameraRay(C.Structure):
    _fields_ = [(n, C.c_float * 3) for n in ("origin", "dir", "weight", "dOdx", "dOdy", "dDdx", "dDdy")]


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    lib = bokeh.load_host_library()
    d, i, vp = C.c_double, C.c_int, C.c_void_p
    pd = C.POINTER(C.c_double)
    pf = C.POINTER(C.c_float)
    pu = C.POINTER(C.c_uint32)
    sig = {
        "lentil_host_lens_create": (vp, [C.POINTER(_abi.LensTable)]),
        "lentil_host_lens_destroy": (None, [vp]),
        "lentil_host_lens_evaluate": (d, [vp, pd, pd]),
        "lentil_host_lens_pt_sample_aperture": (None, [vp, pd, pd, d]),
        "lentil_host_lens_lt_sample_aperture": (d, [vp, pd, pd, pd, pd, d]),
        "lentil_host_camera_get_y0_intersection_distance": (d, [vp, d, d]),
        "lentil_host_logarithmic_focus_search": (d, [vp, d, d]),
        "lentil_host_trace_backwards_for_fstop": (None, [vp, d, d, pd, pd]),
        "lentil_host_trace_ray_focus_check": (i, [vp, d, d, pd]),
        "lentil_host_camera_model_specific_setup": (i, [C.POINTER(_abi.Params), vp, d, d, d, pd]),
        "lentil_host_camera_model_specific_setup_with": (i, [C.POINTER(_abi.Params), vp, d, d, d, pd, vp, vp]),
        "lentil_host_xor128_init": (None, [pu]),
        "lentil_host_trace_ray_fw_po": (None, [C.POINTER(_abi.Params), vp, C.POINTER(_abi.BokehTable), pu, d, d, d, pd, pd, i,
                                               pf, pf, pf, C.POINTER(i)]),
        "lentil_host_trace_ray_fw_thinlens": (None, [C.POINTER(_abi.Params), C.POINTER(_abi.BokehTable), pu, d, d, pd, pd, i,
                                                     pf, pf, pf, C.POINTER(i)]),
        "lentil_host_camera_create_ray": (None, [C.POINTER(_abi.Params), vp, C.POINTER(_abi.BokehTable), pu, d, C.c_float,
                                                 pf, C.POINTER(CameraRay)]),
        "lentil_host_camera_reverse_ray": (None, [d, pf, pf]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class HostLens:
    def __init__(self, spec):
        self.lib = load()
        self.table, self._keep = lens_io.make_lens_table(spec)
        self.h = self.lib.lentil_host_lens_create(C.byref(self.table))
        if not self.h:
            raise ValueError("invalid lens table")

    def close(self):
        if self.h:
            self.lib.lentil_host_lens_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def camera_model_specific_setup(params, lens=None, input_fstop=0.0, wavelength_nm=550.0, extra_sensor_shift=0.0):
    """Camera::camera_model_specific_setup (src/lentil.h:1568-1670) through the host library; returns tan_fov."""
    lib = load()
    tan_fov = C.c_double()
    rc = lib.lentil_host_camera_model_specific_setup(C.byref(params), lens.h if lens else None, input_fstop,
                                                     wavelength_nm, extra_sensor_shift, C.byref(tan_fov))
    if rc:
        raise ValueError("camera_model_specific_setup: invalid arguments")
    return tan_fov.value
