"""Host-side camera setup: produces the lentil_params the kernels consume.

Mirrors (in numpy, fp64, same operation order) the parts of the reference's
Camera::get_lentil_camera_params / camera_model_specific_setup / setup_filter
(src/lentil.h:1189-1243, 1568-1670, 1057-1122) that the redistribution path depends on:
parameter defaults and clamps, the PO aperture radius, the logarithmic focus search that
yields `sensor_shift`, and the render-region / buffer-size quirk (xres = W + 1).
"""
import numpy as np

from . import _abi, lens_io


def _ipow(x, e):
    """lens_ipow, src/lens.h:226-233 (same recursion -> same roundings)."""
    if e == 0:
        return np.ones_like(x)
    if e == 1:
        return x
    if e == 2:
        return x * x
    p2 = _ipow(x, e // 2)
    if e & 1:
        return x * p2 * p2
    return p2 * p2


def poly_eval(terms, v):
    """terms: [(c, [e0..e4])]; v: list of 5 arrays/scalars.  sum of c*f1*f2.. left to right."""
    total = None
    for c, e in terms:
        t = np.float64(c)
        for i in range(5):
            if e[i] == 0:
                continue
            t = t * (v[i] if e[i] == 1 else _ipow(v[i], e[i]))
        total = t if total is None else total + t
    return total


def poly_derive(terms, var):
    return [(float(c) * float(e[var]), [ee - (1 if i == var else 0) for i, ee in enumerate(e)])
            for c, e in terms if e[var] > 0]


def _normalise(v):
    ilen = 1.0 / np.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])
    return [v[0] * ilen, v[1] * ilen, v[2] * ilen]


def sphere_to_cs(inpos, indir, center, R):
    """sphereToCs, src/lens.h:99-125 (vectorised)."""
    n = [inpos[0] / R, inpos[1] / R,
         np.sqrt(np.maximum(0.0, R * R - inpos[0] * inpos[0] - inpos[1] * inpos[1])) / abs(R)]
    t = [indir[0], indir[1], np.sqrt(np.maximum(0.0, 1.0 - indir[0] * indir[0] - indir[1] * indir[1]))]
    ex = _normalise([n[2], np.zeros_like(n[2]), -n[0]])
    ey = [n[1] * ex[2] - n[2] * ex[1], n[2] * ex[0] - n[0] * ex[2], n[0] * ex[1] - n[1] * ex[0]]
    d = [t[0] * ex[i] + t[1] * ey[i] + t[2] * n[i] for i in range(3)]
    return [inpos[0], inpos[1], n[2] * R + center], d


class LensModel:
    """A lens table + the evaluation the reference's setup code needs (a13)."""

    def __init__(self, spec):
        self.spec = lens_io.load_lens_json(spec) if isinstance(spec, str) else spec
        self.k = self.spec["constants"]
        self.polys = self.spec["polys"]
        self.dap = [[poly_derive(self.polys["ap_" + a], 2 + j) for j in range(2)] for a in ("x", "y")]
        self.dap_pos = [[poly_derive(self.polys["ap_" + a], j) for j in range(2)] for a in ("x", "y")]

    def lens_evaluate(self, x, y, dx, dy, lam):
        """Camera::lens_evaluate, src/lentil.h:1257-1266."""
        v = [x, y, dx, dy, lam]
        out = [poly_eval(self.polys[n], v) for n in lens_io.OUT_NAMES]
        return out[:4], np.maximum(0.0, out[4])

    def pt_sample_aperture(self, x, y, dx, dy, lam, out_x, out_y, dist):
        """Camera::lens_pt_sample_aperture, src/lentil.h:1272-1291 (generated body restated:
        <= 5 Newton steps on the aperture position, tolerance 1e-4).  Vectorised: every lane runs
        until its own error is below tolerance."""
        x, y, dx, dy = [np.array(a, np.float64, copy=True) for a in np.broadcast_arrays(x, y, dx, dy)]
        sqr_err = np.full(x.shape, 3.4028234663852886e38)
        pred_dx = np.zeros_like(x)
        pred_dy = np.zeros_like(x)
        for _ in range(5):
            act = sqr_err > 1e-4
            if not act.any():
                break
            v = [x + dist * dx, y + dist * dy, dx, dy, lam]
            px_ = poly_eval(self.polys["ap_x"], v)
            py_ = poly_eval(self.polys["ap_y"], v)
            pdx = poly_eval(self.polys["ap_dx"], v)
            pdy = poly_eval(self.polys["ap_dy"], v)
            J = [[poly_eval(self.dap[i][j], v) + dist * poly_eval(self.dap_pos[i][j], v) for j in range(2)]
                 for i in range(2)]
            invdet = 1.0 / (J[0][0] * J[1][1] - J[0][1] * J[1][0])
            inv = [[J[1][1] * invdet, -J[0][1] * invdet], [-J[1][0] * invdet, J[0][0] * invdet]]
            d = [out_x - px_, out_y - py_]
            ndx, ndy = dx, dy
            for i in range(2):
                ndx = ndx + inv[0][i] * d[i]
                ndy = ndy + inv[1][i] * d[i]
            dx = np.where(act, ndx, dx)
            dy = np.where(act, ndy, dy)
            pred_dx = np.where(act, pdx, pred_dx)
            pred_dy = np.where(act, pdy, pred_dy)
            sqr_err = np.where(act, d[0] * d[0] + d[1] * d[1], sqr_err)
        return dx, dy, pred_dx, pred_dy

    def y0_intersection_distance(self, sensor_shift, lam):
        """Camera::camera_get_y0_intersection_distance, src/lentil.h:1361-1386 (vectorised over shifts)."""
        ss = np.asarray(sensor_shift, np.float64)
        z = np.zeros_like(ss)
        dx, dy, _, _ = self.pt_sample_aperture(z, z, z, z, lam, 0.0, self.k["lens_aperture_housing_radius"] * 0.25, ss)
        sx = z + dx * ss
        sy = z + dy * ss
        out, _ = self.lens_evaluate(sx, sy, dx, dy, lam)
        R = self.k["lens_outer_pupil_curvature_radius"]
        pos, d = sphere_to_cs(out[:2], out[2:], -R, R)
        # line_plane_intersection, src/lens.h:412-419: Eigen's normalize() divides by the norm, and the
        # result is rayOrigin + (rayDirection * (0 - origin.y)) / rayDirection.y
        nrm = np.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])
        d = [d[0] / nrm, d[1] / nrm, d[2] / nrm]
        return pos[2] + (d[2] * (0.0 - pos[1])) / d[1]

    def logarithmic_focus_search(self, focal_distance, lam):
        """Camera::logarithmic_focus_search, src/lentil.h:1445-1460 + logarithmic_values, src/lens.h:395-407."""
        vals = []
        i = -1.0
        while i <= 1.0:
            vals.append((-1 if i < 0 else 1) * (i ** 2.0) * 45.0 + 0.0)
            i += 0.0001
        shifts = np.array(vals, np.float64)
        with np.errstate(all="ignore"):
            dist = self.y0_intersection_distance(shifts, lam)
        new_distance = focal_distance - dist
        closest, best = 999999999.0, 0.0
        for nd, s in zip(new_distance, shifts):
            if nd < closest and nd > 0.0:
                closest, best = nd, s
        return float(best)


def default_params():
    """node_parameters defaults (src/lentil_camera.cpp:19-52) after get_lentil_camera_params()."""
    p = _abi.Params()
    p.cameraType = _abi.THINLENS
    p.unitModel = _abi.UNIT_CM
    p.enable_dof = 1
    p.vignetting_retries = 15
    p.bokeh_aperture_blades = 0
    p.bokeh_enable_image = 0
    p.bidir_sample_mult = 5
    p.enable_bidir_transmission = 0
    p.enable_skydome = 0
    p.abb_chromatic_type = 0
    p.adaptive_sampling = 0
    p.samples_override = 0
    p.sensor_width = 36.0
    p.focus_distance = 150.0
    p.focal_length = 35.0
    p.bidir_add_energy = 0.0
    p.bidir_add_energy_minimum_luminance = 2.0
    p.bidir_add_energy_transition = 1.0
    p.abb_spherical = 0.5
    p.abb_coma = 0.0
    p.abb_distortion = 0.0
    p.abb_chromatic = 0.0
    p.circle_to_square = np.float32(0.01)      # clamp(0, 0.01, 0.99), src/lentil.h:1226-1227
    p.bokeh_anamorphic = 1.0                   # clamp(1 - 0, 0, 1), src/lentil.h:1228-1229
    p.optical_vignetting_distance = 0.0
    p.optical_vignetting_radius = 1.0
    p.filter_width = 1.5
    p.inverse_sample_density = 1.0 / 9.0
    p.lambda_bw = 0.55
    for i in range(4):
        for j in range(4):
            p.world_to_camera[i][j] = 1.0 if i == j else 0.0
    return p


def setup_filter(p, width, height, region=None, filter_width=1.5, aa_samples=3):
    """Camera::setup_filter, src/lentil.h:1057-1122: region defaults and the (W+1, H+1) buffer."""
    p.xres_without_region, p.yres_without_region = width, height
    if region is None:
        rminx, rminy, rmaxx, rmaxy = 0, 0, width, height       # region_max = res when unset (:1073-1076)
    else:
        rminx, rminy, rmaxx, rmaxy = region
    p.region_min_x, p.region_min_y = rminx, rminy
    p.xres = rmaxx - rminx + 1
    p.yres = rmaxy - rminy + 1
    p.filter_width = filter_width
    p.inverse_sample_density = np.float32(1.0) / (np.float32(aa_samples) * np.float32(aa_samples))
    return p


def setup_thinlens(p, focal_length=35.0, fstop=1.4, focus_dist=150.0, sensor_width=36.0):
    """ThinLens branch of camera_model_specific_setup, src/lentil.h:1663-1668 (host library)."""
    from . import hostlib
    p.cameraType = _abi.THINLENS
    p.sensor_width = float(np.float32(sensor_width))
    p.focal_length = max(np.float32(focal_length), np.float32(0.01))        # clamp_min, src/lentil.h:1217
    p.focus_distance = float(np.float32(focus_dist))
    input_fstop = float(max(np.float32(fstop), np.float32(0.01)))           # src/lentil.h:1206
    hostlib.camera_model_specific_setup(p, None, input_fstop)
    return p


def setup_po(p, lens, focus_dist=150.0, sensor_width=36.0, focal_length=None, wavelength_nm=550.0,
             extra_sensor_shift=0.0, fstop=0.0):
    """PolynomialOptics branch of camera_model_specific_setup, src/lentil.h:1571-1662, run by the host
    library (liblentil_host.so): focus_distance cm -> mm, aperture radius (fstop 0 = wide open, else the
    backward f-stop trace), logarithmic focus search for the sensor shift."""
    from . import hostlib
    model = lens if isinstance(lens, LensModel) else LensModel(lens)
    p.cameraType = _abi.POLYNOMIAL_OPTICS
    p.sensor_width = float(np.float32(sensor_width))
    p.focus_distance = float(np.float32(focus_dist))                       # cm; the setup multiplies by 10 (:1573)
    # get_coc_thinlens uses the thin-lens focal length even in PO mode (src/lentil.h:674-692)
    p.focal_length = np.float32(focal_length if focal_length is not None else model.k["lens_effective_focal_length"])
    hl = hostlib.HostLens(model.spec)
    input_fstop = float(max(np.float32(fstop), np.float32(0.01))) if fstop else 0.0
    hostlib.camera_model_specific_setup(p, hl, input_fstop, wavelength_nm, extra_sensor_shift)
    hl.close()
    return p, model
