"""ctypes mirror of include/lentil_hip.h (data layout only -- no compute here)."""
import ctypes as C

LENTIL_MAX_AOVS = 16
LENTIL_MAX_CRYPTO = 9
THINLENS, POLYNOMIAL_OPTICS = 0, 1
UNIT_MM, UNIT_CM, UNIT_DM, UNIT_M = 0, 1, 2, 3
FILTER_GAUSSIAN, FILTER_CLOSEST, FILTER_VARIANCE, FILTER_CLOSEST_DEBUG = 0, 1, 2, 3
GEOM_SPHERICAL, GEOM_CYL_Y, GEOM_CYL_X = 0, 1, 2
OK, ERR_INVALID, ERR_HIP, ERR_UNSUPPORTED, ERR_NOMEM = 0, -1, -2, -3, -4


class Params(C.Structure):
    _fields_ = [
        ("cameraType", C.c_int32), ("unitModel", C.c_int32), ("enable_dof", C.c_int32),
        ("vignetting_retries", C.c_int32), ("bokeh_aperture_blades", C.c_int32),
        ("bokeh_enable_image", C.c_int32), ("bidir_sample_mult", C.c_int32),
        ("enable_bidir_transmission", C.c_int32), ("enable_skydome", C.c_int32),
        ("abb_chromatic_type", C.c_int32), ("adaptive_sampling", C.c_int32),
        ("samples_override", C.c_int32),
        ("xres", C.c_uint32), ("yres", C.c_uint32),
        ("xres_without_region", C.c_uint32), ("yres_without_region", C.c_uint32),
        ("region_min_x", C.c_int32), ("region_min_y", C.c_int32),
        ("sensor_width", C.c_double), ("focus_distance", C.c_double),
        ("aperture_radius", C.c_double), ("sensor_shift", C.c_double),
        ("bidir_add_energy_minimum_luminance", C.c_double),
        ("focal_length", C.c_float), ("bidir_add_energy", C.c_float),
        ("bidir_add_energy_transition", C.c_float), ("abb_spherical", C.c_float),
        ("abb_coma", C.c_float), ("abb_distortion", C.c_float), ("abb_chromatic", C.c_float),
        ("circle_to_square", C.c_float), ("bokeh_anamorphic", C.c_float),
        ("optical_vignetting_distance", C.c_float), ("optical_vignetting_radius", C.c_float),
        ("filter_width", C.c_float), ("inverse_sample_density", C.c_float),
        ("lambda_bw", C.c_float),
        ("world_to_camera", (C.c_float * 4) * 4),
    ]


class Term(C.Structure):
    _fields_ = [("c", C.c_double), ("e", C.c_uint8 * 5), ("pad", C.c_uint8 * 3)]


class Poly(C.Structure):
    _fields_ = [("first", C.c_uint32), ("count", C.c_uint32)]


LENS_CONSTANT_NAMES = [
    "lens_outer_pupil_radius", "lens_inner_pupil_radius", "lens_length",
    "lens_back_focal_length", "lens_effective_focal_length", "lens_aperture_pos",
    "lens_aperture_housing_radius", "lens_inner_pupil_curvature_radius",
    "lens_outer_pupil_curvature_radius", "lens_field_of_view", "lens_fstop",
    "lens_aperture_radius_at_fstop",
]


class LensTable(C.Structure):
    _fields_ = [(n, C.c_double) for n in LENS_CONSTANT_NAMES] + [
        ("lens_inner_pupil_geometry", C.c_int32), ("lens_outer_pupil_geometry", C.c_int32),
        ("out", Poly * 5), ("ap", Poly * 4),
        ("n_terms", C.c_uint32), ("reserved", C.c_uint32),
        ("terms", C.POINTER(Term)),
    ]


class BokehTable(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32),
                ("cdfRow", C.c_void_p), ("rowIndices", C.c_void_p),
                ("cdfColumn", C.c_void_p), ("columnIndices", C.c_void_p)]


class Visits(C.Structure):
    _fields_ = [
        ("n", C.c_uint64), ("visits_per_pixel", C.c_uint32), ("pixels_per_row", C.c_uint32),
        ("pixel_x0", C.c_int32), ("pixel_y0", C.c_int32), ("pixel_row_stride", C.c_uint32),
        ("n_extra", C.c_uint32),
        ("rgba", C.c_void_p), ("pos_z", C.c_void_p), ("raydir_time", C.c_void_p),
        ("volume_ignore", C.c_void_p), ("transmission", C.c_void_p),
        ("extra", C.c_void_p * (LENTIL_MAX_AOVS - 1)),
        ("pixel", C.c_void_p), ("inv_density", C.c_void_p),
    ]


class CryptoVisits(C.Structure):
    _fields_ = [("n", C.c_uint64), ("n_crypto", C.c_uint32), ("entries", C.c_uint32),
                ("hash", C.c_void_p * LENTIL_MAX_CRYPTO), ("weight", C.c_void_p * LENTIL_MAX_CRYPTO)]


class Counters(C.Structure):
    _fields_ = [("visits", C.c_uint64), ("redistributed_visits", C.c_uint64),
                ("attempted_draws", C.c_uint64), ("accepted_draws", C.c_uint64),
                ("worklist_overflow", C.c_uint64), ("newton_iterations", C.c_uint64),
                ("tries", C.c_uint64), ("lane_rounds", C.c_uint64), ("slow_solves", C.c_uint64),
                ("blind_chunks", C.c_uint64), ("fallback_chunks", C.c_uint64), ("streamed", C.c_uint64)]


class PassTotals(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "passes", "streamed", "blind_chunks", "fallback_chunks", "deferred", "abandoned", "abandoned_incomplete",
        "visits", "redistributed_visits", "attempted_draws", "accepted_draws", "worklist_overflow",
        "newton_iterations", "tries", "lane_rounds", "slow_solves", "scan_launches", "rounds_max")] + [
        ("scan_ms", C.c_double), ("draw_ms", C.c_double), ("resolve_ms", C.c_double)]


class DrawRecord(C.Structure):
    _fields_ = [("visit", C.c_uint32), ("attempt", C.c_uint32), ("pixel", C.c_uint32)]
