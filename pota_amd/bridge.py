"""ctypes binding of liblentil_bridge.so (include/lentil_bridge.h): the SDK-free host logic of the
plugin's node callbacks -- camera parameter mapping, filter node facts, operator AOV planning, visit
capture (staging) and the imager's once-only GPU pass + bucket copy."""
import ctypes as C
import os

import numpy as np

from . import _abi, capi

AI_TYPE = {"INT": 0x01, "BOOLEAN": 0x03, "FLOAT": 0x04, "RGB": 0x05, "RGBA": 0x06, "VECTOR": 0x07,
           "STRING": 0x0A, "ENUM": 0x0F, "NONE": 0xFF}


class NodeParam(C.Structure):
    _fields_ = [("name", C.c_char_p), ("type", C.c_int), ("default_value", C.c_double),
                ("default_string", C.c_char_p), ("enum_values", C.POINTER(C.c_char_p))]


class CameraNodeValues(C.Structure):
    _fields_ = [
        ("camera_type", C.c_int), ("bidir_sample_mult", C.c_int), ("units", C.c_int), ("sensor_width", C.c_float),
        ("enable_dof", C.c_int), ("fstop", C.c_float), ("focus_dist", C.c_float), ("aperture_blades_lentil", C.c_int),
        ("exp", C.c_float), ("lens_model", C.c_int), ("wavelength", C.c_float), ("extra_sensor_shift", C.c_float),
        ("focal_length_lentil", C.c_float), ("optical_vignetting", C.c_float), ("abb_spherical", C.c_float),
        ("abb_distortion", C.c_float), ("abb_coma", C.c_float), ("abb_chromatic", C.c_float),
        ("abb_chromatic_type", C.c_int), ("bokeh_circle_to_square", C.c_float), ("bokeh_anamorphic", C.c_float),
        ("bokeh_enable_image", C.c_int), ("bokeh_image_path", C.c_char_p), ("vignetting_retries", C.c_int),
        ("bidir_add_energy", C.c_float), ("bidir_add_energy_minimum_luminance", C.c_float),
        ("bidir_add_energy_transition", C.c_float), ("enable_bidir_transmission", C.c_int), ("enable_skydome", C.c_int),
    ]


class OutputTokens(C.Structure):
    _fields_ = [("camera", C.c_char * 128), ("aov_name", C.c_char * 128), ("aov_type", C.c_char * 32),
                ("filter", C.c_char * 128), ("driver", C.c_char * 128), ("half_flag", C.c_int)]


class AovPlan(C.Structure):
    _fields_ = [("to", OutputTokens), ("name", C.c_char * 128), ("type", C.c_uint), ("original_filter", C.c_int),
                ("is_duplicate", C.c_int), ("is_crypto", C.c_int), ("index", C.c_int)]


class SampleCapture(C.Structure):
    _fields_ = [("px", C.c_int), ("py", C.c_int), ("inverse_sample_density", C.c_float), ("rgba", C.c_float * 4),
                ("P", C.c_float * 3), ("Z", C.c_float), ("raydir", C.c_float * 3), ("time", C.c_float),
                ("volume", C.c_float * 3), ("bidir_ignore", C.c_float), ("transmission", C.c_float * 4),
                ("extra_rgba", C.POINTER(C.c_float)), ("crypto_ids", C.POINTER(C.c_float)),
                ("crypto_weights", C.POINTER(C.c_float))]


EXPORTS = [
    "lentil_camera_node_parameters", "lentil_camera_node_defaults", "lentil_camera_params_from_node",
    "lentil_filter_required_aovs", "lentil_filter_width", "lentil_filter_output_type",
    "lentil_filter_inverse_sample_density", "lentil_tokenize_output", "lentil_rebuild_output",
    "lentil_string_to_arnold_type", "lentil_operator_cook", "lentil_sanitize_aov_list", "lentil_aov_frame_kind",
    "lentil_stage_create", "lentil_stage_destroy", "lentil_stage_reset", "lentil_stage_append", "lentil_stage_size",
    "lentil_stage_visits", "lentil_stage_stream_to", "lentil_stage_finish_stream", "lentil_stage_is_streaming", "lentil_imager_create", "lentil_imager_destroy", "lentil_imager_new_frame",
    "lentil_imager_process_bucket", "lentil_imager_last_error",
    "lentil_setup_crypto_aovs", "lentil_crypto_construct_cache", "lentil_crypto_rank_of_name", "lentil_stage_set_crypto",
    "lentil_stage_crypto", "lentil_imager_set_crypto", "lentil_imager_process_crypto_bucket",
    "lentil_setup_filter_region", "lentil_filter_gaussian_complete", "lentil_filter_closest_complete", "lentil_lens_model_name", "lentil_lens_model_table",
]

_lib = None


def load():
    """Loads liblentil_hip.so first (one HIP runtime per process, see capi.load_library), then the bridge."""
    global _lib
    if _lib is not None:
        return _lib
    capi.load_library()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "liblentil_bridge.so")
    if not os.path.exists(path):
        raise RuntimeError("liblentil_bridge.so is missing: run python -c 'import __graft_entry__ as g; g.build()'")
    lib = C.CDLL(path)
    i, f, d, vp, u32, u64 = C.c_int, C.c_float, C.c_double, C.c_void_p, C.c_uint32, C.c_uint64
    sig = {
        "lentil_camera_node_parameters": (C.POINTER(NodeParam), [C.POINTER(i)]),
        "lentil_camera_node_defaults": (None, [C.POINTER(CameraNodeValues)]),
        "lentil_camera_params_from_node": (i, [C.POINTER(CameraNodeValues), f, i, C.POINTER(_abi.Params), C.POINTER(d),
                                               C.POINTER(d), C.POINTER(d), C.POINTER(f)]),
        "lentil_filter_required_aovs": (C.POINTER(C.c_char_p), []),
        "lentil_filter_width": (f, [i]),
        "lentil_filter_output_type": (i, [i]),
        "lentil_filter_inverse_sample_density": (f, [i, f, i, C.POINTER(i)]),
        "lentil_tokenize_output": (None, [C.c_char_p, C.POINTER(OutputTokens)]),
        "lentil_rebuild_output": (i, [C.POINTER(OutputTokens), C.c_char_p, C.c_size_t]),
        "lentil_string_to_arnold_type": (C.c_uint, [C.c_char_p]),
        "lentil_operator_cook": (i, [C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), i, C.POINTER(AovPlan), i, C.c_char_p,
                                     C.c_size_t]),
        "lentil_sanitize_aov_list": (i, [C.POINTER(AovPlan), i]),
        "lentil_aov_frame_kind": (i, [C.POINTER(AovPlan)]),
        "lentil_stage_create": (i, [i, u32, C.POINTER(vp)]),
        "lentil_stage_destroy": (None, [vp]),
        "lentil_stage_reset": (None, [vp]),
        "lentil_stage_append": (i, [vp, i, C.POINTER(SampleCapture)]),
        "lentil_stage_size": (u64, [vp]),
        "lentil_stage_visits": (i, [vp, C.POINTER(_abi.Visits)]),
        "lentil_stage_stream_to": (i, [vp, vp, u32, u64]),
        "lentil_stage_finish_stream": (i, [vp, C.POINTER(u64)]),
        "lentil_stage_is_streaming": (i, [vp]),
        "lentil_imager_create": (i, [vp, vp, C.POINTER(_abi.Params), u32, C.POINTER(vp)]),
        "lentil_imager_destroy": (None, [vp]),
        "lentil_imager_new_frame": (None, [vp]),
        "lentil_imager_process_bucket": (i, [vp, u32, i, i, i, i, vp]),
        "lentil_setup_crypto_aovs": (i, [C.POINTER(C.c_char_p), i, C.POINTER(AovPlan), i]),
        "lentil_crypto_construct_cache": (i, [i, vp, vp, vp, vp, i]),
        "lentil_crypto_rank_of_name": (i, [C.c_char_p]),
        "lentil_stage_set_crypto": (i, [vp, u32, u32]),
        "lentil_stage_crypto": (i, [vp, C.POINTER(_abi.CryptoVisits)]),
        "lentil_imager_set_crypto": (i, [vp, u32, C.POINTER(i)]),
        "lentil_imager_process_crypto_bucket": (i, [vp, u32, i, i, i, i, vp]),
        "lentil_imager_last_error": (C.c_char_p, [vp]),
        "lentil_setup_filter_region": (None, [C.POINTER(_abi.Params), i, i, i, i, i, i, f]),
        "lentil_filter_gaussian_complete": (None, [i, vp, vp, vp, f, f, vp, vp]),
        "lentil_filter_closest_complete": (None, [i, vp, vp, vp]),
        "lentil_lens_model_name": (C.c_char_p, [i]),
        "lentil_lens_model_table": (C.c_char_p, [i]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def camera_node_parameters():
    """The camera node's parameter table (src/lentil_camera.cpp:19-52) as a list of dicts."""
    lib = load()
    n = C.c_int()
    arr = lib.lentil_camera_node_parameters(C.byref(n))
    out = []
    for k in range(n.value):
        p = arr[k]
        enum = []
        if p.enum_values:
            j = 0
            while p.enum_values[j]:
                enum.append(p.enum_values[j].decode())
                j += 1
        out.append({"name": p.name.decode(), "type": int(p.type), "default": float(p.default_value),
                    "default_string": p.default_string.decode() if p.default_string else None, "enum_values": enum})
    return out


def operator_cook(outputs, filter_entry_names):
    """Returns (plans, warnings): plans is a list of AovPlan copies."""
    lib = load()
    n = len(outputs)
    arr_o = (C.c_char_p * n)(*[o.encode() for o in outputs])
    arr_f = (C.c_char_p * n)(*[x.encode() for x in filter_entry_names])
    cap = n + 3
    plans = (AovPlan * cap)()
    warn = C.create_string_buffer(4096)
    m = lib.lentil_operator_cook(arr_o, arr_f, n, plans, cap, warn, len(warn))
    if m < 0:
        raise ValueError("lentil_operator_cook failed")
    return [plans[k] for k in range(m)], warn.value.decode()


def rebuild_output(tok):
    buf = C.create_string_buffer(1024)
    n = load().lentil_rebuild_output(C.byref(tok), buf, len(buf))
    if n < 0:
        raise ValueError("buffer too small")
    return buf.value.decode()


def stage_append_arrays(stage, slot, cols, idx):
    """Test helper: append visits idx of a column dict (as produced by pota_amd.workload) one by one,
    the way filter_pixel would."""
    lib = load()
    n_extra = len(cols.get("extra", []))
    n_crypto = len(cols.get("crypto_ids", []))
    for v in idx:
        c = SampleCapture()
        px = int(cols["pixel"][v])
        c.px, c.py = px & 0xFFFF, px >> 16
        c.inverse_sample_density = float(cols["inv_density"][v])
        c.rgba[:] = cols["rgba"][v]
        c.P[:] = cols["pos_z"][v, :3]; c.Z = cols["pos_z"][v, 3]
        c.raydir[:] = cols["raydir_time"][v, :3]; c.time = cols["raydir_time"][v, 3]
        c.volume[:] = cols["volume_ignore"][v, :3]; c.bidir_ignore = cols["volume_ignore"][v, 3]
        c.transmission[:] = cols["transmission"][v]
        if n_extra:
            ex = np.ascontiguousarray(np.stack([cols["extra"][k][v] for k in range(n_extra)]).astype(np.float32))
            c.extra_rgba = ex.ctypes.data_as(C.POINTER(C.c_float))
        if n_crypto:       # per cryptomatte AOV the sample's cache, entries (id, weight) pairs
            ci = np.ascontiguousarray(np.stack([cols["crypto_ids"][k][v] for k in range(n_crypto)]).astype(np.float32))
            cw = np.ascontiguousarray(np.stack([cols["crypto_weights"][k][v] for k in range(n_crypto)]).astype(np.float32))
            c.crypto_ids = ci.ctypes.data_as(C.POINTER(C.c_float))
            c.crypto_weights = cw.ctypes.data_as(C.POINTER(C.c_float))
        rc = lib.lentil_stage_append(stage, slot, C.byref(c))
        if rc:
            raise RuntimeError("lentil_stage_append rc=%d" % rc)
