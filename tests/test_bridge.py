"""Host logic of the plugin's node callbacks (include/lentil_bridge.h, liblentil_bridge.so): camera
parameter table and mapping, filter node facts, output-string tokens, the operator's AOV list, visit
capture -- CPU tests; the imager's once-only GPU pass and bucket copy is a GPU test."""
import ctypes as C
import os
import re
import threading

import numpy as np
import pytest

import common
from pota_amd import _abi, bridge, camera, capi, workload


def test_header_symbols_are_exported():
    lib = bridge.load()
    txt = open(os.path.join(common.ROOT, "include", "lentil_bridge.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = sorted(set(re.findall(r"\b(lentil_(?!hip_)\w+)\s*\(", txt)))
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "liblentil_bridge.so does not export %s" % n
    assert sorted(bridge.EXPORTS) == names


def test_camera_node_parameters_match_the_reference_declarations():
    """node_parameters of lentil_camera, src/lentil_camera.cpp:20-49: 29 parameters, order, types, defaults."""
    lib = bridge.load()
    n = C.c_int()
    p = lib.lentil_camera_node_parameters(C.byref(n))
    assert n.value == 29
    names = [p[i].name.decode() for i in range(n.value)]
    assert names[:5] == ["camera_type", "bidir_sample_mult", "units", "sensor_width", "enable_dof"]
    assert names[-2:] == ["enable_bidir_transmission", "enable_skydome"]
    d = {p[i].name.decode(): p[i] for i in range(n.value)}
    assert d["focus_dist"].default_value == 150.0 and d["vignetting_retries"].default_value == 15
    assert d["units"].type == bridge.AI_TYPE["ENUM"] and d["units"].default_value == 1     # cm
    assert [d["units"].enum_values[k] for k in range(5)] == [b"mm", b"cm", b"dm", b"m", b"automatic"]
    assert d["bokeh_image_path"].type == bridge.AI_TYPE["STRING"] and d["bokeh_image_path"].default_string == b""
    assert d["abb_spherical"].default_value == 0.5 and d["focal_length_lentil"].default_value == 35.0


def test_default_node_values_give_the_default_params():
    """get_lentil_camera_params (src/lentil.h:1189-1243) on the node defaults == camera.default_params()."""
    lib = bridge.load()
    v = bridge.CameraNodeValues()
    lib.lentil_camera_node_defaults(C.byref(v))
    out = camera.default_params()
    want = camera.default_params()
    fstop, lam, shift, exp = C.c_double(), C.c_double(), C.c_double(), C.c_float()
    assert lib.lentil_camera_params_from_node(C.byref(v), 0.01, 0, C.byref(out), C.byref(fstop), C.byref(lam),
                                              C.byref(shift), C.byref(exp)) == 0
    for name, _ in _abi.Params._fields_:
        a, b = getattr(out, name), getattr(want, name)
        if hasattr(a, "_length_"):
            continue
        assert a == b, name
    assert fstop.value == float(np.float32(0.01))            # clamp_min(0.0, 0.01)
    assert abs(lam.value - 0.55) < 1e-12 and exp.value == 1.0


def test_node_value_clamps_and_the_automatic_unit_quirk():
    lib = bridge.load()
    v = bridge.CameraNodeValues()
    lib.lentil_camera_node_defaults(C.byref(v))
    v.abb_spherical = 2.0; v.bokeh_circle_to_square = 0.0; v.bokeh_anamorphic = 0.25; v.focal_length_lentil = 0.0
    v.units = 4
    out = camera.default_params()
    args = (C.byref(out), None, None, None, None)
    lib.lentil_camera_params_from_node(C.byref(v), 1.0, 1, *args)
    assert out.unitModel == 3 and out.enable_dof == 0                    # metres; options.ignore_dof wins
    assert out.abb_spherical == np.float32(0.999) and out.circle_to_square == np.float32(0.01)
    assert out.bokeh_anamorphic == np.float32(0.75) and out.focal_length == np.float32(0.01)
    # the float option is compared with double literals: 0.01f != 0.01, "automatic" stays unresolved
    lib.lentil_camera_params_from_node(C.byref(v), 0.01, 0, *args)
    assert out.unitModel == 4


def test_filter_node_facts(orc):
    lib = bridge.load()
    req = lib.lentil_filter_required_aovs()
    got = []
    while req[len(got)]:
        got.append(req[len(got)].decode())
    assert got == ["RGBA RGBA", "VECTOR P", "FLOAT Z", "FLOAT lentil_time", "FLOAT lentil_debug", "RGB lentil_raydir",
                   "RGB opacity", "RGBA transmission", "FLOAT lentil_bidir_ignore"]
    assert lib.lentil_filter_width(0) == 1.5 and lib.lentil_filter_width(1) == 1.0
    T = bridge.AI_TYPE
    for t in ("RGBA", "RGB", "VECTOR", "FLOAT"):
        assert lib.lentil_filter_output_type(T[t]) == T["RGBA"]
    assert lib.lentil_filter_output_type(T["INT"]) == T["NONE"]
    # visit prologue vs the oracle's restatement of src/lentil_filter.cpp:79-88
    for count, width, aa in [(9, 1.0, 3), (36, 1.5, 4), (20, 1.5, 3), (4, 1.0, 2), (81, 1.5, 6), (16, 1.0, 4)]:
        dis, ok = C.c_int(), C.c_int()
        a = lib.lentil_filter_inverse_sample_density(count, width, aa, C.byref(dis))
        b = orc.orc_inverse_sample_density(count, width, aa, C.byref(ok))
        assert a == b and dis.value == (0 if ok.value else 1)


def test_output_tokens_round_trip():
    """TokenizedOutputLentil, src/aov_data.h:30-91"""
    lib = bridge.load()
    cases = {
        "RGBA RGBA gaussian_filter driver_exr": ("", "RGBA", "RGBA", "gaussian_filter", "driver_exr", 0),
        "persp RGBA RGBA gaussian_filter driver_exr": ("persp", "RGBA", "RGBA", "gaussian_filter", "driver_exr", 0),
        "Z FLOAT closest_filter drv HALF": ("", "Z", "FLOAT", "closest_filter", "drv", 1),
        "cam N VECTOR closest_filter drv HALF": ("cam", "N", "VECTOR", "closest_filter", "drv", 1),
        "too few": ("", "", "", "", "", 0),
    }
    for s, want in cases.items():
        t = bridge.OutputTokens()
        lib.lentil_tokenize_output(s.encode(), C.byref(t))
        got = (t.camera.decode(), t.aov_name.decode(), t.aov_type.decode(), t.filter.decode(), t.driver.decode(), t.half_flag)
        assert got == want, s
        if want[1]:
            assert bridge.rebuild_output(t) == s
    for s, ty in [("FLOAT", "FLOAT"), ("flt", "FLOAT"), ("rgba", "RGBA"), ("RGB", "RGB"), ("vec", "VECTOR")]:
        assert lib.lentil_string_to_arnold_type(s.encode()) == bridge.AI_TYPE[ty]
    assert lib.lentil_string_to_arnold_type(b"INT") == 0


def test_operator_cook_plans_the_aov_list():
    """operator_cook, src/lentil_operator.cpp:41-127 + sanitize_aov_list, src/aov_data.h:164-174"""
    outputs = [
        "RGBA RGBA gaussian_filter drv",
        "P VECTOR closest_filter drv",
        "albedo RGB blackman_harris_filter drv",       # incompatible filter -> gaussian + warning
        "ID UINT closest_filter drv",                   # unsupported type: keeps its filter
        "crypto_material RGB gaussian_filter drv",      # display-only cryptomatte layer: keeps its filter
        "crypto_material00 RGBA gaussian_filter drv",   # ranked cryptomatte layer: skipped here
        "RGBA RGBA gaussian_filter drv2",               # second driver: duplicate
        "N VECTOR variance_filter drv HALF",
    ]
    entries = ["gaussian_filter", "closest_filter", "blackman_harris_filter", "closest_filter", "gaussian_filter",
               "gaussian_filter", "gaussian_filter", "variance_filter"]
    plans, warn = bridge.operator_cook(outputs, entries)
    names = [a.name.decode() for a in plans]
    assert names == ["RGBA", "P", "albedo", "ID", "crypto_material", "RGBA", "N", "lentil_debug", "lentil_time", "lentil_raydir"]
    assert "blackman_harris_filter" in warn and warn.count("[LENTIL]") == 1
    by = {(a.name.decode(), a.to.driver.decode()): a for a in plans}
    assert by[("RGBA", "drv")].to.filter == b"lentil_replaced_filter" and by[("RGBA", "drv")].is_duplicate == 0
    assert by[("RGBA", "drv2")].is_duplicate == 1
    assert by[("P", "drv")].original_filter == _abi.FILTER_CLOSEST
    assert by[("albedo", "drv")].original_filter == _abi.FILTER_GAUSSIAN
    assert by[("ID", "drv")].to.filter == b"closest_filter" and by[("ID", "drv")].type == 0
    assert by[("crypto_material", "drv")].to.filter == b"gaussian_filter"
    assert by[("N", "drv")].original_filter == _abi.FILTER_VARIANCE and by[("N", "drv")].to.half_flag == 1
    dbg = by[("lentil_debug", "drv")]
    assert dbg.original_filter == _abi.FILTER_CLOSEST and dbg.type == bridge.AI_TYPE["FLOAT"]
    assert bridge.rebuild_output(dbg.to) == "lentil_debug FLOAT lentil_replaced_filter drv"
    assert bridge.rebuild_output(by[("lentil_raydir", "drv")].to) == "lentil_raydir RGB lentil_replaced_filter drv"
    # setup_filter's list: duplicates and AOVs lentil does not filter are dropped, the rest numbered
    arr = (bridge.AovPlan * len(plans))(*plans)
    m = bridge.load().lentil_sanitize_aov_list(arr, len(plans))
    kinds = [bridge.load().lentil_aov_frame_kind(C.byref(arr[k])) for k in range(m)]
    assert kinds == [_abi.FILTER_GAUSSIAN, _abi.FILTER_CLOSEST, _abi.FILTER_GAUSSIAN, _abi.FILTER_VARIANCE,
                     _abi.FILTER_CLOSEST_DEBUG, _abi.FILTER_GAUSSIAN, _abi.FILTER_GAUSSIAN]
    kept = [(arr[k].name.decode(), arr[k].index) for k in range(m)]
    assert kept == [("RGBA", 0), ("P", 1), ("albedo", 2), ("N", 3), ("lentil_debug", 4), ("lentil_time", 5), ("lentil_raydir", 6)]


def _ragged_columns(W, H, M, f_hi, n_extra, p, seed=3):
    n = W * H * M
    cols = workload.generate(np, 0, n, W, H, M, f_hi=f_hi, focus_dist=150.0, tan_half_fov=common.tan_half_fov(p),
                             n_extra=n_extra)
    pix = np.arange(n, dtype=np.uint32) // M
    cols["pixel"] = ((pix % W) | ((pix // W) << 16)).astype(np.uint32)
    cols["inv_density"] = np.full(n, p.inverse_sample_density, np.float32)
    return cols


def test_visit_capture_from_concurrent_threads():
    """filter_pixel is re-entrant: several render threads append to their own slot; the concatenated
    stream holds every visit exactly once, slot by slot in append order."""
    lib = bridge.load()
    p, model, table, keep = common.po_setup(24, 16)
    cols = _ragged_columns(24, 16, 9, 0.05, 2, p)
    n = cols["rgba"].shape[0]
    stage = C.c_void_p()
    assert lib.lentil_stage_create(4, 2, C.byref(stage)) == 0
    parts = np.array_split(np.arange(n), 4)
    th = [threading.Thread(target=bridge.stage_append_arrays, args=(stage, k, cols, parts[k])) for k in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert lib.lentil_stage_size(stage) == n
    v = _abi.Visits()
    assert lib.lentil_stage_visits(stage, C.byref(v)) == 0
    assert v.n == n and v.visits_per_pixel == 0 and v.n_extra == 2
    def col(ptr, width, dtype=np.float32):
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float if dtype == np.float32 else C.c_uint32)), (n * width,)).reshape(n, width) if width > 1 \
            else np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float if dtype == np.float32 else C.c_uint32)), (n,))
    for name in ("rgba", "pos_z", "raydir_time", "volume_ignore", "transmission"):
        assert np.array_equal(col(getattr(v, name), 4), cols[name]), name
    for k in range(2):
        assert np.array_equal(col(v.extra[k], 4), cols["extra"][k])
    assert np.array_equal(col(v.pixel, 1, np.uint32), cols["pixel"])
    assert np.array_equal(col(v.inv_density, 1), cols["inv_density"])
    # invalid arguments are errors, not crashes
    c = bridge.SampleCapture()
    assert lib.lentil_stage_append(stage, 7, C.byref(c)) == _abi.ERR_INVALID
    assert lib.lentil_stage_append(stage, 0, C.byref(c)) == _abi.ERR_INVALID       # extra AOVs expected
    lib.lentil_stage_reset(stage)
    assert lib.lentil_stage_size(stage) == 0
    lib.lentil_stage_destroy(stage)


@pytest.mark.gpu
@pytest.mark.parametrize("streaming", [False, True], ids=["staged", "streamed-upload"])
def test_imager_buckets_match_the_direct_pipeline(orc, gpu_ctx_factory, streaming):
    """driver_process_bucket (src/lentil_imager.cpp:66-193): buckets requested from several threads; the
    GPU pass runs once; the assembled frame equals the one the C-ABI pipeline delivers directly.
    "streamed-upload": the stage sends its visits to the GPU in page-locked blocks while they are appended
    (lentil_stage_stream_to; blocks of 1000 visits so that every slot sends several and reuses its two blocks),
    then a second frame through the same stage."""
    lib = bridge.load()
    W, H, M = 80, 48, 9
    kinds = [0, 1, 0]
    p, model, table, keep = common.po_setup(W, H, samples_override=32)
    cols = _ragged_columns(W, H, M, 0.02, 2, p)
    n = cols["rgba"].shape[0]
    # the frame the plain C-ABI sequence produces
    direct = gpu_ctx_factory()
    visits, kv = capi.make_visits(cols, visits_per_pixel=0)
    direct.set_params(p); direct.set_lens(table); direct.set_bokeh(None); direct.alloc_frame(3, kinds)
    direct.upload_visits(visits); direct.clear_frame(); direct.redistribute(); direct.resolve()
    want = [direct.download_aov(a).reshape(p.yres, p.xres, 4) for a in range(3)]
    n_draws = direct.counters().accepted_draws
    assert n_draws > 1000
    # ... which is the oracle's frame (so the buckets below are compared with the reference's restatement, not only with
    # another run of the same kernels): accepted draws bit-identical, accumulators and resolved AOVs at 1e-5
    import oracle_lib
    from test_gpu_parity import check_frame
    lens = orc.orc_lens_create(C.byref(table))
    ref = oracle_lib.Frame(orc, p, n_aovs=3, kinds=kinds, keep_log=True)
    ref.run(lens, None, visits)
    orc.orc_lens_destroy(lens)
    direct.sync()
    check_frame(direct, ref, n_aovs=3, kinds=kinds)
    ref.close()

    ctx = gpu_ctx_factory()
    ctx.set_params(p); ctx.set_lens(table); ctx.set_bokeh(None); ctx.alloc_frame(3, kinds)
    stage = C.c_void_p()
    assert lib.lentil_stage_create(3, 2, C.byref(stage)) == 0
    order = np.arange(n).reshape(-1, M)            # whole pixels per thread, like buckets
    parts = np.array_split(order, 3)
    im = C.c_void_p()
    assert lib.lentil_imager_create(ctx.h, stage, C.byref(p), 3, C.byref(im)) == 0
    if streaming:
        assert lib.lentil_stage_stream_to(stage, ctx.h, 1000, 0) == 0
        assert lib.lentil_stage_is_streaming(stage) == 1
        # a first frame with other content through the same stage: its blocks and device columns are reused
        for k in range(3):
            bridge.stage_append_arrays(stage, k, cols, parts[(k + 1) % 3].reshape(-1)[: 5000 + 700 * k])
        assert lib.lentil_stage_size(stage) == 3 * 5000 + 700 * 3
        buf = np.empty((4, 4, 4), np.float32)
        assert lib.lentil_imager_process_bucket(im, 0, 0, 0, 4, 4, buf.ctypes.data) == 0
        lib.lentil_stage_reset(stage)
        lib.lentil_imager_new_frame(im)
        th = [threading.Thread(target=bridge.stage_append_arrays, args=(stage, k, cols, parts[k].reshape(-1))) for k in range(3)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert lib.lentil_stage_size(stage) == n
    else:
        for k in range(3):
            bridge.stage_append_arrays(stage, k, cols, parts[k].reshape(-1))

    B = 16
    got = [np.full((p.yres, p.xres, 4), np.nan, np.float32) for _ in range(3)]
    buckets = [(x, y) for y in range(0, p.yres, B) for x in range(0, p.xres, B)]
    errs = []

    def worker(my):
        for (x, y) in my:
            sx, sy = min(B, p.xres - x), min(B, p.yres - y)
            for a in range(3):
                buf = np.empty((sy, sx, 4), np.float32)
                rc = lib.lentil_imager_process_bucket(im, a, x, y, sx, sy, buf.ctypes.data)
                if rc:
                    errs.append(lib.lentil_imager_last_error(im))
                got[a][y:y + sy, x:x + sx] = buf

    th = [threading.Thread(target=worker, args=(buckets[k::4],)) for k in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    assert ctx.counters().accepted_draws == n_draws        # one pass, same draws
    assert np.array_equal(got[1], want[1])                  # closest AOV: exact
    for a in (0, 2):
        m = want[a] != 0
        assert np.array_equal(got[a] != 0, m)
        assert float(np.max(np.abs(got[a][m] - want[a][m]) / np.abs(want[a][m]))) < 1e-5
    # a bucket that hangs over the frame edge leaves the outside untouched
    buf = np.full((8, 8, 4), -7.0, np.float32)
    assert lib.lentil_imager_process_bucket(im, 0, p.xres - 4, p.yres - 4, 8, 8, buf.ctypes.data) == 0
    assert np.all(buf[4:, :] == -7.0) and np.all(buf[:, 4:] == -7.0) and np.array_equal(buf[:4, :4], got[0][-4:, -4:])
    lib.lentil_imager_destroy(im)
    lib.lentil_stage_destroy(stage)


def test_display_pass_through_equals_the_oracle(orc):
    """a21: lentil_filter_gaussian_complete / _closest_complete (what lentil.so's filter_pixel returns for display)
    against the oracle's restatement of Camera::filter_gaussian_complete / filter_closest_complete
    (src/lentil.h:696-775), bit for bit: samples outside the filter radius, non-positive densities, adaptive and uniform
    densities, equal and zero depths, libm's exp and an SDK-style fast exp handed in by the caller."""
    lib = bridge.load()
    fake = C.CDLL(os.path.join(common.ROOT, "tests", "fake_arnold", "libai_fake.so"))
    fast_exp = C.cast(getattr(fake, "_Z9AiFastExpf"), C.c_void_p)          # (C++ linkage, like the rest of the stand-in SDK)
    rng = np.random.default_rng(21)
    RGBA, RGB, FLOAT, VECTOR = 6, 5, 4, 7
    differs = 0
    for case in range(200):
        n = int(rng.integers(0, 40))
        off = rng.uniform(-0.9, 0.9, (n, 2)).astype(np.float32)
        val = rng.uniform(-2, 5, (n, 4)).astype(np.float32)
        dens = rng.choice(np.array([1 / 9, 1 / 16, 0.0, -1.0, 0.3], np.float32), n).astype(np.float32)
        fw = np.float32(rng.choice([1.0, 1.5, 2.0]))
        adaptive = int(rng.integers(0, 2))
        uniform = np.float32(1 / 9)
        for typ in (RGBA, RGB):
            v = val.copy()
            if typ == RGB:
                v[:, 3] = 1.0                   # what the plugin hands over for an RGB AOV (AtRGB -> AtRGBA)
            for fe in (None, fast_exp):
                got = np.zeros(4, np.float32); want = np.zeros(4, np.float32)
                d_arr = dens if adaptive else np.full(n, uniform, np.float32)
                lib.lentil_filter_gaussian_complete(n, off.ctypes.data, v.ctypes.data, d_arr.ctypes.data, C.c_float(0.0),
                                                    C.c_float(fw), fe, got.ctypes.data)
                orc.orc_filter_gaussian_complete(n, off.ctypes.data, val.ctypes.data, dens.ctypes.data, typ, C.c_float(uniform),
                                                 adaptive, C.c_float(fw), fe, want.ctypes.data)
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (case, typ, fe is not None)
            a = np.zeros(4, np.float32); b = np.zeros(4, np.float32)
            d_arr = dens if adaptive else np.full(n, uniform, np.float32)
            lib.lentil_filter_gaussian_complete(n, off.ctypes.data, v.ctypes.data, d_arr.ctypes.data, C.c_float(0.0), C.c_float(fw), None, a.ctypes.data)
            lib.lentil_filter_gaussian_complete(n, off.ctypes.data, v.ctypes.data, d_arr.ctypes.data, C.c_float(0.0), C.c_float(fw), fast_exp, b.ctypes.data)
            differs += int(not np.array_equal(a, b))
        depth = rng.choice(np.array([0.0, 1.0, -1.0, 2.5, 7.0, -7.0], np.float32), n).astype(np.float32)
        for typ in (VECTOR, FLOAT):
            v = val.copy()
            if typ == FLOAT:
                v[:, 1] = v[:, 0]; v[:, 2] = v[:, 0]
            v[:, 3] = 1.0
            got = np.zeros(4, np.float32); want = np.zeros(4, np.float32)
            lib.lentil_filter_closest_complete(n, depth.ctypes.data, v.ctypes.data, got.ctypes.data)
            orc.orc_filter_closest_complete(n, depth.ctypes.data, val.ctypes.data, typ, want.ctypes.data)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (case, typ)
    assert differs > 50            # the caller's exp function is the one that is used


def test_lens_model_enum_is_the_references():
    """lens_model keeps the reference's ids and numbers (include/auto_generated_lens_includes/pota_h_lenses.h:4-47; default
    cooke__speed_panchro__1920__40mm, src/lentil_camera.cpp:29): a scene written for the reference parses.  Tables ship for
    two of the ids (stand-ins), the rest are refused by name at camera update."""
    lib = bridge.load()
    d = {w["name"]: w for w in bridge.camera_node_parameters()}["lens_model"]
    ids = d["enum_values"]
    assert len(ids) == 47 and ids[0] == "angenieux__double_gauss__1953__49mm" and ids[43] == "zeiss__biotar__1927__45mm"
    assert ids[int(d["default"])] == "cooke__speed_panchro__1920__40mm" and int(d["default"]) == 16
    assert ids[25] == "kodak__petzval__1948__58mm" and ids[20] == "kodak__petzval__1948__150mm"
    assert [i for i in ids[:44] if i.count("__") != 3] == []            # maker__design__year__focal-length
    assert ids[44:] == ["double_gauss_50mm", "petzval_58mm", "anamorphic_petzval_58mm"]
    assert lib.lentil_lens_model_name(0) == b"angenieux__double_gauss__1953__49mm"
    assert lib.lentil_lens_model_table(0) == b"double_gauss_50mm" and lib.lentil_lens_model_table(25) == b"petzval_58mm"
    assert lib.lentil_lens_model_table(16) is None and lib.lentil_lens_model_table(46) == b"anamorphic_petzval_58mm"
    assert lib.lentil_lens_model_name(47) is None and lib.lentil_lens_model_name(-1) is None
