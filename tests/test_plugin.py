"""lentil.so -- the Arnold plugin DSO -- driven by a stand-in renderer (tests/fake_arnold, test infrastructure: a fake
<ai.h> plus libai_fake.so, which also plays Arnold: node registration through NodeLoader, operator cook, node update,
filter_pixel from several threads, driver_process_bucket per bucket).

CPU tests: the registration surface (four nodes in the reference's order, src/lentil_loader.cpp:11-28), the camera
node's parameters against the reference's list (src/lentil_camera.cpp:19-52 as lentil_camera_node_parameters() states
it), node metadata, what lentil_operator leaves in options.outputs / aov_shaders (src/lentil_operator.cpp:25-171).
GPU test: a whole frame through the plugin -- operator_cook -> node_update -> filter_pixel on 6 threads ->
driver_process_bucket on 6 threads -- against the ORACLE (not against the direct HIP path).
"""
import ctypes as C
import os

import numpy as np
import pytest

import common
import oracle_lib
from pota_amd import _abi, bridge, capi, workload

FAKE = os.path.join(common.ROOT, "tests", "fake_arnold", "libai_fake.so")
PLUGIN = os.path.join(common.ROOT, "pota_amd", "lentil.so")
AI_NODE_CAMERA, AI_NODE_DRIVER, AI_NODE_FILTER, AI_NODE_OPERATOR = 0x0002, 0x0040, 0x0080, 0x1000
AI_TYPE = {"INT": 1, "BOOLEAN": 3, "FLOAT": 4, "RGB": 5, "RGBA": 6, "VECTOR": 7, "STRING": 10, "ENUM": 15}


@pytest.fixture(scope="module")
def fa():
    lib = C.CDLL(FAKE, mode=C.RTLD_GLOBAL)
    lib.fa_entry_name.restype = C.c_char_p
    lib.fa_entry_meta.restype = C.c_char_p
    lib.fa_output.restype = C.c_char_p
    lib.fa_universe_create.restype = C.c_void_p
    lib.fa_node.restype = C.c_void_p
    lib.fa_options.restype = C.c_void_p
    for f in ("fa_node", "fa_set_camera", "fa_node_set_int", "fa_node_set_flt", "fa_node_set_bool", "fa_node_set_str",
              "fa_add_output", "fa_output_count", "fa_output", "fa_set_samples", "fa_set_aov", "fa_render", "fa_get_image",
              "fa_cook_operators", "fa_universe_destroy", "fa_node_exists", "fa_aov_shader_count", "fa_render_hint",
              "fa_filter_width_x1000", "fa_options", "fa_set_depths", "fa_set_depth_aov", "fa_add_late_output",
              "fa_add_aov_shader", "fa_get_display_image", "fa_camera_create_ray", "fa_camera_reverse_ray", "fa_camera_set_matrix_keys"):
        getattr(lib, f).argtypes = None
    assert lib.fa_load_plugin(PLUGIN.encode()) == 4
    return lib


def _messages(fa):
    buf = C.create_string_buffer(1 << 16)
    fa.fa_messages(buf, len(buf))
    return buf.value.decode()


def test_node_loader_registers_the_four_nodes(fa):
    names = [fa.fa_entry_name(i).decode() for i in range(fa.fa_entry_count())]
    assert names == ["lentil_camera", "lentil_filter", "imager_lentil", "lentil_operator"]
    assert [fa.fa_entry_node_type(i) for i in range(4)] == [AI_NODE_CAMERA, AI_NODE_FILTER, AI_NODE_DRIVER, AI_NODE_OPERATOR]
    for n in names:
        assert fa.fa_entry_meta(n.encode(), None, b"ai_version").decode().startswith("7.")


def test_camera_parameters_match_the_reference_list(fa):
    want = bridge.camera_node_parameters()
    assert fa.fa_entry_param_count(b"lentil_camera") == len(want) == 29
    name = C.create_string_buffer(128)
    s = C.create_string_buffer(4096)
    typ, num = C.c_int(), C.c_double()
    for i, w in enumerate(want):
        assert fa.fa_entry_param(b"lentil_camera", i, name, 128, C.byref(typ), C.byref(num), s, 4096) == 0
        assert name.value.decode() == w["name"]
        assert typ.value == w["type"]
        if w["type"] == AI_TYPE["STRING"]:
            assert s.value.decode() == (w["default_string"] or "")
        elif w["type"] == AI_TYPE["ENUM"]:
            assert s.value.decode().split("|") == w["enum_values"] and int(num.value) == int(w["default"])
        else:
            assert num.value == pytest.approx(w["default"])
    assert fa.fa_entry_meta(b"lentil_camera", None, b"force_update") == b"true"


def test_node_metadata_and_imager_parameters(fa):
    assert fa.fa_entry_meta(b"imager_lentil", None, b"subtype") == b"imager"
    assert fa.fa_entry_meta(b"lentil_filter", None, b"force_update") == b"true"
    assert fa.fa_entry_meta(b"lentil_operator", None, b"force_update") == b"true"
    assert fa.fa_entry_param_count(b"imager_lentil") == 1
    name, s = C.create_string_buffer(64), C.create_string_buffer(64)
    typ, num = C.c_int(), C.c_double()
    fa.fa_entry_param(b"imager_lentil", 0, name, 64, C.byref(typ), C.byref(num), s, 64)
    assert (name.value, typ.value, num.value) == (b"enable", AI_TYPE["BOOLEAN"], 1.0)
    assert fa.fa_entry_param_count(b"lentil_filter") == 0 and fa.fa_entry_param_count(b"lentil_operator") == 0


def _scene(fa, W, H, outputs, aa=3, oidn=True):
    u = fa.fa_universe_create(W, H, aa)
    cam = fa.fa_node(C.c_void_p(u), b"lentil_camera", b"camera")
    fa.fa_set_camera(C.c_void_p(u), C.c_void_p(cam))
    for f in ("gaussian_filter", "closest_filter", "box_filter"):
        fa.fa_node(C.c_void_p(u), f.encode(), f.encode())
    fa.fa_node(C.c_void_p(u), b"driver_exr", b"driver_exr")
    if oidn:
        fa.fa_node(C.c_void_p(u), b"imager_denoiser_oidn", b"oidn")       # filter width 1.0: a pixel's own samples
    fa.fa_node(C.c_void_p(u), b"imager_lentil", b"imager_lentil")
    fa.fa_node(C.c_void_p(u), b"lentil_operator", b"lentil_operator")
    for o in outputs:
        fa.fa_add_output(C.c_void_p(u), o.encode())
    return u, cam


def test_operator_cook_rewires_the_outputs(fa):
    fa.fa_messages_clear()
    u, cam = _scene(fa, 32, 24, ["RGBA RGBA gaussian_filter driver_exr", "diffuse RGB gaussian_filter driver_exr",
                                 "N VECTOR closest_filter driver_exr", "Z FLOAT box_filter driver_exr",
                                 "ID UINT closest_filter driver_exr"])
    assert fa.fa_cook_operators(C.c_void_p(u)) == 1
    outs = [fa.fa_output(C.c_void_p(u), i).decode() for i in range(fa.fa_output_count(C.c_void_p(u)))]
    assert outs == ["RGBA RGBA lentil_replaced_filter driver_exr", "diffuse RGB lentil_replaced_filter driver_exr",
                    "N VECTOR lentil_replaced_filter driver_exr", "Z FLOAT lentil_replaced_filter driver_exr",
                    "ID UINT closest_filter driver_exr",
                    "lentil_debug FLOAT lentil_replaced_filter driver_exr", "lentil_time FLOAT lentil_replaced_filter driver_exr",
                    "lentil_raydir RGB lentil_replaced_filter driver_exr"]
    for n in (b"lentil_replaced_filter", b"lentil_time_write", b"lentil_time_read", b"lentil_raydir_write", b"lentil_raydir_read"):
        assert fa.fa_node_exists(C.c_void_p(u), n) == 1
    assert fa.fa_aov_shader_count(C.c_void_p(u)) == 2
    assert "Specified AOV filter (box_filter) is incompatible with Lentil" in _messages(fa)
    # a second cook (Arnold re-cooks on scene edits) adds nothing twice
    assert fa.fa_cook_operators(C.c_void_p(u)) == 1
    assert fa.fa_aov_shader_count(C.c_void_p(u)) == 2
    fa.fa_universe_destroy(C.c_void_p(u))


def test_a_reference_lens_id_without_a_table_is_refused_by_name(fa):
    """A scene written for the reference: lens_model is one of its 44 ids (here the node's default,
    cooke__speed_panchro__1920__40mm, src/lentil_camera.cpp:29).  The id parses; in polynomial-optics mode the camera
    update says which table is missing and aborts the render (AiMsgError + AiRenderAbort, like src/lentil.h:225-228)."""
    fa.fa_messages_clear()
    u, cam = _scene(fa, 32, 24, ["RGBA RGBA gaussian_filter driver_exr"])
    fa.fa_node_set_int(C.c_void_p(cam), b"camera_type", 1)                 # PolynomialOptics, lens_model left at its default
    assert fa.fa_render(C.c_void_p(u), 2, 16) == -1
    assert "no polynomial table shipped for lens_model cooke__speed_panchro__1920__40mm (16)" in _messages(fa)
    fa.fa_universe_destroy(C.c_void_p(u))
    fa.fa_messages_clear()


def test_shipped_mtd_equals_the_references_in_everything_a_translator_acts_on():
    """pota_amd/plugin/lentil.mtd (tools/gen_mtd.py) against the reference's own lentil.mtd (tests/golden/lentil.mtd:
    generated here from src/lentil_camera.ui by the reference's uigen.py and concatenated with src/lentil_hardcode.mtd
    like src/CMakeLists.txt:47-67 does; tools/make_mtd_fixture.py): the same nodes, node-level keys, attributes, and per
    attribute the same keys with the same types and VALUES -- labels, ranges, linkability, disable rules, Maya node ids,
    hidden imager attributes, the operator's block commented out.  Only the free-text descriptions are this repo's own."""
    import re
    FREE_TEXT = ("desc", "houdini.help")

    def parse(path):
        nodes, node, attr, pending = {}, None, None, None
        for line in open(path):
            t = line.strip()
            if not t or t.startswith("#"):
                continue
            m = re.match(r"\[node (\w+)\]", t)
            if m:
                node = nodes.setdefault(m.group(1), {"keys": {}, "attrs": {}}); attr = None; pending = None
                continue
            m = re.match(r"\[attr (\w+)\]", t)
            if m:
                attr = node["attrs"].setdefault(m.group(1), {}); pending = None
                continue
            m = re.match(r"([\w\.]+)\s+(\w+)\s+(.*)$", t)
            if m:
                tgt = attr if attr is not None else node["keys"]
                tgt[m.group(1)] = [m.group(2), m.group(3).strip()]
                pending = tgt[m.group(1)]
            elif t.startswith('"') and pending is not None:       # a string continued on the next line (houdini.order)
                pending[1] += " " + t
        return nodes
    ours = parse(os.path.join(common.ROOT, "pota_amd", "plugin", "lentil.mtd"))
    ref = parse(os.path.join(common.ROOT, "tests", "golden", "lentil.mtd"))
    want = [w["name"] for w in bridge.camera_node_parameters()]
    assert list(ours["lentil_camera"]["attrs"]) == want          # declaration order
    assert sorted(ref["lentil_camera"]["attrs"]) == sorted(want)  # (the .ui lists them by UI group)
    assert sorted(ours) == sorted(ref) == ["imager_lentil", "lentil_camera", "lentil_filter"]
    n_values = 0
    for node in ref:
        strip = lambda d: {k: v for k, v in d.items() if k not in FREE_TEXT}
        assert strip(ours[node]["keys"]) == strip(ref[node]["keys"]), node
        assert sorted(ours[node]["attrs"]) == sorted(ref[node]["attrs"]), node
        for a, kv in ref[node]["attrs"].items():
            assert strip(ours[node]["attrs"][a]) == strip(kv), (node, a)
            assert set(ours[node]["attrs"][a]) == set(kv), (node, a)           # (the free-text keys exist on both sides)
            n_values += len(strip(kv))
    assert n_values > 150
    assert ours["imager_lentil"]["attrs"] == {"layer_selection": {"maya.hide": ["BOOL", "false"]}, "input": {"maya.hide": ["BOOL", "TRUE"]}}


@pytest.mark.gpu
def test_frame_through_the_plugin_matches_the_oracle(fa, orc, monkeypatch):
    W, H, M, S = 64, 48, 9, 48
    monkeypatch.setenv("LENTIL_SAMPLES_OVERRIDE", str(S))
    fa.fa_messages_clear()
    u, cam = _scene(fa, W, H, ["RGBA RGBA gaussian_filter driver_exr", "diffuse RGB gaussian_filter driver_exr"])
    fa.fa_node_set_int(C.c_void_p(cam), b"camera_type", 1)                 # PolynomialOptics
    fa.fa_node_set_int(C.c_void_p(cam), b"lens_model", 0)                  # double_gauss_50mm
    # the same frame for the oracle: parameters as the plugin derives them
    # (focal_length: the node's focal_length_lentil default, which the reference's CoC formula uses in PO mode too,
    # src/lentil.h:674-692,1217)
    p, model, table, keep = common.po_setup(W, H, samples_override=S, focal_length=np.float32(35.0))
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.02, n_extra=1)
    n = W * H * M
    pix = np.arange(n) // M
    px = (pix % W).astype(np.int32); py = (pix // W).astype(np.int32)
    rng = np.random.default_rng(5)
    ox = rng.uniform(-0.5, 0.5, n).astype(np.float32); oy = rng.uniform(-0.5, 0.5, n).astype(np.float32)
    invd = np.full(n, 1.0 / 9.0, np.float32)
    fa.fa_set_samples(C.c_void_p(u), n, px.ctypes.data_as(C.c_void_p), py.ctypes.data_as(C.c_void_p), ox.ctypes.data_as(C.c_void_p),
                      oy.ctypes.data_as(C.c_void_p), invd.ctypes.data_as(C.c_void_p))
    zeros = np.zeros((n, 4), np.float32)
    z4 = np.repeat(cols["pos_z"][:, 3:4], 4, axis=1).copy()
    aov = {"RGBA": (AI_TYPE["RGBA"], cols["rgba"]), "P": (AI_TYPE["VECTOR"], cols["pos_z"]), "Z": (AI_TYPE["FLOAT"], z4),
           "lentil_raydir": (AI_TYPE["RGB"], cols["raydir_time"]), "lentil_time": (AI_TYPE["FLOAT"], zeros),
           "volume": (AI_TYPE["RGB"], zeros), "transmission": (AI_TYPE["RGBA"], zeros), "lentil_ignore": (AI_TYPE["FLOAT"], zeros),
           "diffuse": (AI_TYPE["RGB"], cols["extra"][0])}
    keep_arrays = []
    for name, (t, a) in aov.items():
        a = np.ascontiguousarray(a, np.float32); keep_arrays.append(a)
        fa.fa_set_aov(C.c_void_p(u), name.encode(), t, a.ctypes.data_as(C.c_void_p))
    rc = fa.fa_render(C.c_void_p(u), 6, 16)
    msgs = _messages(fa)
    assert rc == 0 and fa.fa_error_count() == 0, msgs
    assert fa.fa_filter_width_x1000(C.c_void_p(u), b"lentil_replaced_filter") == 1000
    assert fa.fa_render_hint(C.c_void_p(u), b"imager_schedule") == 2 and fa.fa_render_hint(C.c_void_p(u), b"imager_padding") == 0
    assert "Adding aov RGBA" in msgs and "Adding aov diffuse" in msgs
    img = {}
    for name in ("RGBA", "diffuse", "lentil_debug", "lentil_raydir"):
        a = np.zeros((H, W, 4), np.float32)
        assert fa.fa_get_image(C.c_void_p(u), name.encode(), a.ctypes.data_as(C.c_void_p)) == 0
        img[name] = a

    # ---- oracle: AOVs in the plugin's order -- RGBA, diffuse (RGB widened with alpha 1), lentil_debug (own z-buffer,
    # no column), lentil_raydir (RGB widened)
    kinds = [_abi.FILTER_GAUSSIAN, _abi.FILTER_GAUSSIAN, _abi.FILTER_CLOSEST_DEBUG, _abi.FILTER_GAUSSIAN]
    ocols = dict(cols)
    widen = lambda a: np.ascontiguousarray(np.concatenate([a[:, :3], np.ones((n, 1), np.float32)], 1), np.float32)
    ocols["extra"] = [widen(cols["extra"][0]), np.zeros_like(cols["rgba"]), widen(cols["raydir_time"])]
    ovisits, okeep = capi.make_visits(ocols, visits_per_pixel=M, pixels_per_row=W)
    lens = orc.orc_lens_create(C.byref(table))
    ref = oracle_lib.Frame(orc, p, n_aovs=4, kinds=kinds, keep_log=True)
    ref.run(lens, None, ovisits)
    orc.orc_lens_destroy(lens)
    assert ref.counters().redistributed_visits > 300
    untouched = np.ones(p.xres * p.yres, bool)
    untouched[ref.log()[:, 2]] = False
    untouched = untouched.reshape(p.yres, p.xres)[:H, :W]
    assert untouched.sum() > 200
    for k, name in enumerate(("RGBA", "diffuse", "lentil_debug", "lentil_raydir")):
        want = ref.resolve(k).reshape(p.yres, p.xres, 4)[:H, :W]
        got = img[name]
        if kinds[k] == _abi.FILTER_CLOSEST_DEBUG:
            assert np.array_equal(got, want), name
            continue
        m = want != 0
        assert np.array_equal(got != 0, m), name
        err = float(np.max(np.abs(got[m].astype(np.float64) - want[m]) / np.abs(want[m])))
        assert err < 1e-5, (name, err)
        # the capture arrives in thread order, a pixel's samples as a run in iterator order: scan_runs_kernel adds a run up
        # in that order, so a pixel no draw lands on holds the reference's sequential sum (src/lentil.h:938-955) bit for bit
        assert np.array_equal(got[untouched].view(np.uint32), want[untouched].view(np.uint32)), name
    ref.close()

    # ---- a21: what filter_pixel itself returned (the display pass-through, before the imager overwrote the buckets):
    # Camera::filter_gaussian_complete for the RGBA / RGB outputs, filter_closest_complete for the FLOAT one
    # (src/lentil_filter.cpp:453-479, src/lentil.h:696-775), with the stand-in SDK's AiFastExp -- bit for bit
    fast_exp = C.cast(getattr(fa, "_Z9AiFastExpf"), C.c_void_p)
    off = np.ascontiguousarray(np.stack([ox, oy], 1))
    zf = np.ascontiguousarray(z4[:, 0])
    shown = {"RGBA": (AI_TYPE["RGBA"], np.ascontiguousarray(cols["rgba"])),
             "diffuse": (AI_TYPE["RGB"], np.ascontiguousarray(cols["extra"][0])),
             "lentil_raydir": (AI_TYPE["RGB"], np.ascontiguousarray(cols["raydir_time"])),
             "lentil_debug": (AI_TYPE["FLOAT"], np.zeros((n, 4), np.float32))}
    for name, (typ, vals) in shown.items():
        got = np.zeros((H, W, 4), np.float32)
        assert fa.fa_get_display_image(C.c_void_p(u), name.encode(), got.ctypes.data_as(C.c_void_p)) == 0
        want = np.zeros((H, W, 4), np.float32)
        out = np.zeros(4, np.float32)
        for q in range(W * H):
            a, b = q * M, (q + 1) * M
            if typ == AI_TYPE["FLOAT"]:
                orc.orc_filter_closest_complete(M, zf[a:b].ctypes.data, vals[a:b].ctypes.data, typ, out.ctypes.data)
            else:
                # (the sample count -- hence the inverse density -- is only taken for the RGBA AOV, src/lentil_filter.cpp:72-86:
                # without adaptive sampling every other gaussian output is weighted 0 here and shows black until the imager's
                # buckets arrive; the plugin keeps that)
                orc.orc_filter_gaussian_complete(M, off[a:b].ctypes.data, vals[a:b].ctypes.data, invd[a:b].ctypes.data, typ,
                                                 C.c_float(1.0 / 9.0 if name == "RGBA" else 0.0), 0, C.c_float(1.0), fast_exp,
                                                 out.ctypes.data)
            nc = 4 if typ == AI_TYPE["RGBA"] else (1 if typ == AI_TYPE["FLOAT"] else 3)
            want[q // W, q % W, :nc] = out[:nc]
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), name
        if name == "RGBA":
            assert float(np.abs(got).max()) > 0.1

    # ---- f2: camera_create_ray / camera_reverse_ray through lentil.so's method table (src/lentil_camera.cpp:78-172)
    # against the oracle's forward trace (Camera::trace_ray_fw_po, src/lentil.h:283-427) and the finite differences of
    # :95-118.  Lens points near the axis: no vignetting retry, so the thread's xor128 stream is not consulted.
    import math
    lens2 = orc.orc_lens_create(C.byref(table))
    lam = float(np.float32(550.0)) * 0.001
    step = np.float32(0.001)
    inv_step = np.float32(1.0) / step                        # AtVector / float: a multiplication by 1 / f (SDK, recalled)
    rg = np.random.default_rng(3)
    n_checked = 0
    for _ in range(200):
        sx, sy = np.float32(rg.uniform(-0.8, 0.8)), np.float32(rg.uniform(-0.4, 0.4))
        dsx, dsy = np.float32(rg.uniform(0.5, 2.0)), np.float32(rg.uniform(0.5, 2.0))
        lx, ly = np.float32(rg.uniform(0.3, 0.7)), np.float32(rg.uniform(0.3, 0.7))
        inp = (C.c_float * 7)(sx, sy, dsx, dsy, lx, ly, 0.0)
        outp = (C.c_float * 21)()
        assert fa.fa_camera_create_ray(C.c_void_p(u), inp, outp, 0) == 0
        got = np.array(list(outp), np.float32)

        def trace(tsx, tsy, deriv, r):
            st = (C.c_uint32 * 4)(); orc.orc_xor128_init(st)
            o, d, w = (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 3)(1, 1, 1)
            tr = C.c_int()
            orc.orc_trace_ray_fw_po(C.byref(p), lens2, None, st, lam, float(tsx), float(tsy), C.byref(r[0]), C.byref(r[1]), deriv, o, d, w, C.byref(tr))
            return np.array(list(o), np.float32), np.array(list(d), np.float32), np.array(list(w), np.float32), tr.value
        r = (C.c_double(float(lx)), C.c_double(float(ly)))
        o0, d0, w0, t0 = trace(sx, sy, 0, r)
        if t0:                      # a retry would draw from the thread's generator: not this check's subject
            continue
        sxd, syd = np.float32(sx + dsx * step), np.float32(sy + dsy * step)
        o1, d1, _, _ = trace(sxd, sy, 1, r)
        o2, d2, _, _ = trace(sx, syd, 1, r)
        want = np.concatenate([o0, d0, (o1 - o0) * inv_step, (o2 - o0) * inv_step, (d1 - d0) * inv_step, (d2 - d0) * inv_step,
                               w0 * np.float32(1.0)]).astype(np.float32)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (sx, sy, got, want)
        n_checked += 1
    assert n_checked > 150
    orc.orc_lens_destroy(lens2)
    tan_fov = math.tan(model.spec["constants"]["lens_field_of_view"] / 2.0)
    for po in ((10.0, 5.0, -100.0), (-3.0, 0.5, -0.001), (0.25, -0.75, 40.0)):
        ps = (C.c_float * 2)()
        assert fa.fa_camera_reverse_ray(C.c_void_p(u), (C.c_float * 3)(*po), C.c_float(0.0), ps) == 1
        coeff = 1.0 / max(abs(float(np.float32(po[2])) * tan_fov), 1e-3)
        assert ps[0] == np.float32(float(np.float32(po[0])) * coeff) and ps[1] == np.float32(float(np.float32(po[1])) * coeff)
    fa.fa_universe_destroy(C.c_void_p(u))


def _fake_matrix_at(keys, time, motion):
    """tests/fake_arnold's AiWorldToCameraMatrix in numpy, operation for operation (fp32)"""
    f32 = np.float32
    ms, me = (f32(motion[0]), f32(motion[1])) if motion else (f32(0), f32(1))
    t = (f32(time) - ms) / (me - ms)
    t = f32(min(max(t, f32(0)), f32(1)))
    n = keys.shape[0]
    sc = f32(t * f32(n - 1))
    i0 = min(int(sc), n - 2)
    f = f32(sc - f32(i0))
    return ((keys[i0 + 1] - keys[i0]) * f + keys[i0]).astype(np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("shutter,motion,n_cam_keys", [((0.0, 1.0), None, 3), ((-0.25, 0.25), (-0.5, 0.5), 3), ((0.0, 0.5), None, 3),
                                                       ((-0.25, 0.25), (0.0, 1.0), 2)],
                         ids=["0..1", "centred", "0..0.5", "knot-inside-the-shutter"])
def test_moving_camera_through_the_plugin_matches_the_oracle(fa, orc, monkeypatch, shutter, motion, n_cam_keys):
    """The camera has matrix keys and every AOV sample its own lentil_time: the reference takes the sample to camera space
    with AiWorldToCameraMatrix(camera, time) (src/lentil_filter.cpp:141-144), `time` being Arnold's absolute sample time
    inside the camera's shutter.  lentil.so samples the matrix over [shutter_start, shutter_end] at camera update, hands
    keys and interval to the GPU path, filter_pixel captures the times; the frame equals the oracle's on the same samples of
    the matrix.  "centred": shutter -0.25 ... 0.25 on a camera whose keys span the motion range -0.5 ... 0.5 -- negative
    times, keys outside the shutter (a 0 ... 1 assumption collapses these onto key 0).
    "knot-inside-the-shutter" (ADVICE round 4): two keys over the motion range 0 ... 1 under a centred shutter -- the camera
    stands still before time 0 and moves after it, a knot that no equidistant sample over the shutter hits; the sixteen
    samples the plugin takes keep the interpolated matrix within a sixteenth of the shutter's motion of the renderer's."""
    W, H, M, S = 48, 32, 9, 32
    monkeypatch.setenv("LENTIL_SAMPLES_OVERRIDE", str(S))
    fa.fa_messages_clear()
    u, cam = _scene(fa, W, H, ["RGBA RGBA gaussian_filter driver_exr"])
    fa.fa_node_set_int(C.c_void_p(cam), b"camera_type", 1)
    fa.fa_node_set_int(C.c_void_p(cam), b"lens_model", 0)
    keys = np.stack([np.eye(4, dtype=np.float32) for _ in range(n_cam_keys)])
    if n_cam_keys == 3:
        keys[1, 3, 0], keys[2, 3, 0], keys[2, 3, 1] = 6.0, 15.0, -4.0
    else:
        keys[1, 3, 0], keys[1, 3, 1] = 15.0, -4.0
    assert fa.fa_camera_set_matrix_keys(C.c_void_p(u), n_cam_keys, keys.ctypes.data_as(C.c_void_p)) == 0
    fa.fa_node_set_flt(C.c_void_p(cam), b"shutter_start", C.c_float(shutter[0]))
    fa.fa_node_set_flt(C.c_void_p(cam), b"shutter_end", C.c_float(shutter[1]))
    if motion:
        fa.fa_node_set_flt(C.c_void_p(cam), b"motion_start", C.c_float(motion[0]))
        fa.fa_node_set_flt(C.c_void_p(cam), b"motion_end", C.c_float(motion[1]))
    # what the plugin samples: the renderer's matrix at LENTIL_MAX_MOTION_KEYS equidistant times over the shutter
    f32 = np.float32
    NS = 16
    sampled = np.stack([_fake_matrix_at(keys, f32(shutter[0]) + (f32(k) / f32(NS - 1)) * (f32(shutter[1]) - f32(shutter[0])), motion)
                        for k in range(NS)])
    # ... and how far the blend of those samples is from the renderer's own matrix at any time of the shutter: nothing
    # where the camera's knots lie on samples or outside the shutter, at most one sample interval's motion at a knot between
    tt = np.linspace(shutter[0], shutter[1], 997).astype(np.float32)
    sc = (tt - f32(shutter[0])) / (f32(shutter[1]) - f32(shutter[0])) * f32(NS - 1)
    i0 = np.minimum(sc.astype(np.int64), NS - 2)
    blend = sampled[i0] + (sampled[i0 + 1] - sampled[i0]) * (sc - i0)[:, None, None]
    true = np.stack([_fake_matrix_at(keys, t, motion) for t in tt])
    moved = float(np.abs(sampled[-1] - sampled[0]).max())
    assert float(np.abs(blend - true).max()) <= moved / (NS - 1) + 1e-4
    p, model, table, keep = common.po_setup(W, H, samples_override=S, focal_length=np.float32(35.0))
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.03)
    n = W * H * M
    times = np.random.default_rng(2).uniform(shutter[0], shutter[1], n).astype(np.float32)
    cols["raydir_time"] = cols["raydir_time"].copy()
    cols["raydir_time"][:, 3] = times
    pix = np.arange(n) // M
    px = (pix % W).astype(np.int32); py = (pix // W).astype(np.int32)
    rng = np.random.default_rng(5)
    ox = rng.uniform(-0.5, 0.5, n).astype(np.float32); oy = rng.uniform(-0.5, 0.5, n).astype(np.float32)
    invd = np.full(n, 1.0 / 9.0, np.float32)
    fa.fa_set_samples(C.c_void_p(u), n, px.ctypes.data_as(C.c_void_p), py.ctypes.data_as(C.c_void_p), ox.ctypes.data_as(C.c_void_p),
                      oy.ctypes.data_as(C.c_void_p), invd.ctypes.data_as(C.c_void_p))
    zeros = np.zeros((n, 4), np.float32)
    z4 = np.repeat(cols["pos_z"][:, 3:4], 4, axis=1).copy()
    t4 = np.repeat(times[:, None], 4, axis=1).copy()
    aov = {"RGBA": (AI_TYPE["RGBA"], cols["rgba"]), "P": (AI_TYPE["VECTOR"], cols["pos_z"]), "Z": (AI_TYPE["FLOAT"], z4),
           "lentil_raydir": (AI_TYPE["RGB"], cols["raydir_time"]), "lentil_time": (AI_TYPE["FLOAT"], t4),
           "volume": (AI_TYPE["RGB"], zeros), "transmission": (AI_TYPE["RGBA"], zeros), "lentil_ignore": (AI_TYPE["FLOAT"], zeros)}
    keep_arrays = []
    for name, (t, a) in aov.items():
        a = np.ascontiguousarray(a, np.float32); keep_arrays.append(a)
        fa.fa_set_aov(C.c_void_p(u), name.encode(), t, a.ctypes.data_as(C.c_void_p))
    rc = fa.fa_render(C.c_void_p(u), 4, 16)
    assert rc == 0 and fa.fa_error_count() == 0, _messages(fa)
    got = np.zeros((H, W, 4), np.float32)
    assert fa.fa_get_image(C.c_void_p(u), b"RGBA", got.ctypes.data_as(C.c_void_p)) == 0
    # the oracle: RGBA, lentil_debug (no column), lentil_raydir -- the plugin's AOV order for this scene
    kinds = [_abi.FILTER_GAUSSIAN, _abi.FILTER_CLOSEST_DEBUG, _abi.FILTER_GAUSSIAN]
    ocols = dict(cols)
    widen = lambda a: np.ascontiguousarray(np.concatenate([a[:, :3], np.ones((n, 1), np.float32)], 1), np.float32)
    ocols["extra"] = [np.zeros_like(cols["rgba"]), widen(cols["raydir_time"])]
    ovisits, okeep = capi.make_visits(ocols, visits_per_pixel=M, pixels_per_row=W)
    lens = orc.orc_lens_create(C.byref(table))
    ref = oracle_lib.Frame(orc, p, n_aovs=3, kinds=kinds)
    ref.set_camera_motion(sampled)
    ref.set_camera_shutter(*shutter)
    ref.run(lens, None, ovisits)
    still = oracle_lib.Frame(orc, p, n_aovs=3, kinds=kinds)
    still.run(lens, None, ovisits)
    orc.orc_lens_destroy(lens)
    want = ref.resolve(0).reshape(p.yres, p.xres, 4)[:H, :W]
    other = still.resolve(0).reshape(p.yres, p.xres, 4)[:H, :W]
    m = want != 0
    assert np.array_equal(got != 0, m)
    assert float(np.max(np.abs(got[m].astype(np.float64) - want[m]) / np.abs(want[m]))) < 1e-5
    assert not np.array_equal(other != 0, m) or float(np.max(np.abs(other[m] - want[m]))) > 1e-2       # the keys matter
    ref.close(); still.close()
    fa.fa_universe_destroy(C.c_void_p(u))


@pytest.mark.gpu
def test_cryptomatte_frame_through_the_plugin_matches_the_oracle(fa, orc, monkeypatch):
    """A scene with a cryptomatte node: its ranked outputs (added after the operator cooked, like cryptomatte's own
    update does) get lentil's filter (Camera::setup_crypto_aovs, src/lentil.h:1015-1055); filter_pixel folds every AOV
    sample's depth entries into an id -> weight cache (cryptomatte_construct_cache, :781-811); the imager writes the
    ranked pairs (src/lentil_imager.cpp:121-161).  Against the oracle run on the same samples with caches built by the
    oracle's own restatement of the cache construction."""
    W, H, M, S = 48, 32, 9, 48
    monkeypatch.setenv("LENTIL_SAMPLES_OVERRIDE", str(S))
    monkeypatch.setenv("LENTIL_CRYPTO_ENTRIES", "4")
    fa.fa_messages_clear()
    u, cam = _scene(fa, W, H, ["RGBA RGBA gaussian_filter driver_exr", "crypto_object RGB gaussian_filter driver_exr"])
    cm = fa.fa_node(C.c_void_p(u), b"cryptomatte", b"cryptomatte1")
    fa.fa_add_aov_shader(C.c_void_p(u), C.c_void_p(cm))
    fa.fa_node(C.c_void_p(u), b"cryptomatte_filter", b"crypto_object_filter00")
    names = ["crypto_object00", "crypto_object01"]
    for nm in names:
        fa.fa_add_late_output(C.c_void_p(u), (nm + " FLOAT crypto_object_filter00 driver_exr").encode())
    fa.fa_node_set_int(C.c_void_p(cam), b"camera_type", 1)
    fa.fa_node_set_int(C.c_void_p(cam), b"lens_model", 0)
    p, model, table, keep = common.po_setup(W, H, samples_override=S, focal_length=np.float32(35.0))
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.03)
    n = W * H * M
    pix = np.arange(n) // M
    px = (pix % W).astype(np.int32); py = (pix // W).astype(np.int32)
    rng = np.random.default_rng(9)
    ox = rng.uniform(-0.5, 0.5, n).astype(np.float32); oy = rng.uniform(-0.5, 0.5, n).astype(np.float32)
    invd = np.full(n, 1.0 / 9.0, np.float32)
    fa.fa_set_samples(C.c_void_p(u), n, px.ctypes.data_as(C.c_void_p), py.ctypes.data_as(C.c_void_p), ox.ctypes.data_as(C.c_void_p),
                      oy.ctypes.data_as(C.c_void_p), invd.ctypes.data_as(C.c_void_p))
    zeros = np.zeros((n, 4), np.float32)
    z4 = np.repeat(cols["pos_z"][:, 3:4], 4, axis=1).copy()
    aov = {"RGBA": (AI_TYPE["RGBA"], cols["rgba"]), "P": (AI_TYPE["VECTOR"], cols["pos_z"]), "Z": (AI_TYPE["FLOAT"], z4),
           "lentil_raydir": (AI_TYPE["RGB"], cols["raydir_time"]), "lentil_time": (AI_TYPE["FLOAT"], zeros),
           "volume": (AI_TYPE["RGB"], zeros), "transmission": (AI_TYPE["RGBA"], zeros), "lentil_ignore": (AI_TYPE["FLOAT"], zeros)}
    keep_arrays = []
    for name, (t, a) in aov.items():
        a = np.ascontiguousarray(a, np.float32); keep_arrays.append(a)
        fa.fa_set_aov(C.c_void_p(u), name.encode(), t, a.ctypes.data_as(C.c_void_p))
    # depth entries: 0..3 per sample; ids follow 6-pixel columns so that neighbouring pixels share objects
    counts = rng.integers(0, 4, n).astype(np.int32)
    total = int(counts.sum())
    start = np.concatenate([[0], np.cumsum(counts)[:-1]])
    owner = np.repeat(np.arange(n), counts)
    opacity = rng.random((total, 1)).astype(np.float32).repeat(4, 1)
    opacity[rng.random(total) < 0.35] = 1.0
    pal = (rng.standard_normal(64) * 50).astype(np.float32)
    idv = pal[((px[owner] // 6) * 3 + rng.integers(0, 3, total)) % 64]
    id4 = np.ascontiguousarray(np.repeat(idv[:, None], 4, 1), np.float32)
    fa.fa_set_depths(C.c_void_p(u), counts.ctypes.data_as(C.c_void_p))
    fa.fa_set_depth_aov(C.c_void_p(u), b"opacity", np.ascontiguousarray(opacity).ctypes.data_as(C.c_void_p))
    for nm in names:
        fa.fa_set_depth_aov(C.c_void_p(u), nm.encode(), id4.ctypes.data_as(C.c_void_p))

    rc = fa.fa_render(C.c_void_p(u), 4, 16)
    msgs = _messages(fa)
    assert rc == 0 and fa.fa_error_count() == 0, msgs
    outs = [fa.fa_output(C.c_void_p(u), i).decode() for i in range(fa.fa_output_count(C.c_void_p(u)))]
    assert "crypto_object00 FLOAT lentil_replaced_filter driver_exr" in outs
    assert "crypto_object01 FLOAT lentil_replaced_filter driver_exr" in outs
    assert "crypto_object RGB gaussian_filter driver_exr" in outs           # the display AOV keeps its filter
    assert "Adding aov crypto_object00" in msgs and "Adding aov crypto_object01" in msgs

    # ---- oracle: caches from the oracle's own cache construction
    entries = 4
    ids = np.zeros((n, entries), np.float32)
    wts = np.full((n, entries), np.array([0xFFFFFFFF], np.uint32).view(np.float32)[0], np.float32)
    op3 = np.ascontiguousarray(opacity[:, :3])
    for v in range(n):
        a, b = int(start[v]), int(start[v]) + int(counts[v])
        o = np.ascontiguousarray(op3[a:b]); val = np.ascontiguousarray(idv[a:b])
        ti = np.empty(entries, np.float32); tw = np.empty(entries, np.float32)
        k = orc.orc_crypto_construct_cache(b - a, o.ctypes.data, val.ctypes.data, ti.ctypes.data, tw.ctypes.data, entries)
        assert 1 <= k <= entries
        ids[v, :k] = ti[:k]; wts[v, :k] = tw[:k]
    lens = orc.orc_lens_create(C.byref(table))
    ref = oracle_lib.Frame(orc, p, n_aovs=1)
    ref.set_crypto([ids, ids], [wts, wts])
    ref.run(lens, None, visits)
    orc.orc_lens_destroy(lens)
    assert ref.counters().redistributed_visits > 200
    B = 16
    for a, nm in enumerate(names):
        rank = 2 * a
        got = np.zeros((H, W, 4), np.float32)
        assert fa.fa_get_image(C.c_void_p(u), nm.encode(), got.ctypes.data_as(C.c_void_p)) == 0
        rout, rhas = ref.crypto_rank(a, rank)
        rout = rout.reshape(p.yres, p.xres, 4)[:H, :W]; rhas = rhas.reshape(p.yres, p.xres)[:H, :W]
        checked = left = 0
        for y in range(H):
            for x0 in range(0, W, B):
                stop = B
                miss = np.nonzero(~rhas[y, x0:x0 + B])[0]
                if len(miss):
                    stop = int(miss[0])
                # the rest of the bucket row keeps the display value: these AOVs have no sample-level data -> 0
                assert (got[y, x0 + stop:x0 + B] == 0).all()
                left += B - stop
                for x in range(x0, x0 + stop):
                    k, w, tot = ref.crypto_pixel(a, y * p.xres + x)
                    ws = np.sort(w)[::-1] / max(tot, 1e-30)
                    lo, hi = max(rank - 1, 0), min(rank + 2, len(ws) - 1)
                    gaps = np.abs(np.diff(ws[lo:hi + 1]))
                    if len(gaps) and gaps.min() < 1e-4:
                        continue
                    assert got[y, x, 0] == rout[y, x, 0] and got[y, x, 2] == rout[y, x, 2], (nm, x, y)
                    assert abs(got[y, x, 1] - rout[y, x, 1]) < 1e-4 and abs(got[y, x, 3] - rout[y, x, 3]) < 1e-4
                    checked += 1
        assert checked > 300, (nm, checked, left)
    # the beauty of the same pass
    got = np.zeros((H, W, 4), np.float32)
    fa.fa_get_image(C.c_void_p(u), b"RGBA", got.ctypes.data_as(C.c_void_p))
    want = ref.resolve(0).reshape(p.yres, p.xres, 4)[:H, :W]
    m = want != 0
    assert float(np.max(np.abs(got[m].astype(np.float64) - want[m]) / np.abs(want[m]))) < 1e-5
    ref.close()
    fa.fa_universe_destroy(C.c_void_p(u))


@pytest.mark.gpu
def test_scene_occlusion_through_the_plugin_matches_the_oracle(fa, orc, monkeypatch):
    """The renderer's scene holds an occluder (the stand-in SDK's analytic sphere behind AiTraceProbe): the camera node hands
    the library a probe callback that asks the renderer the reference's question per try -- AiMakeRay(AI_RAY_SHADOW, sample,
    normalize(lens point - sample), distance) and AiTraceProbe (src/lentil.h:613-629) -- and the frame that comes out of the
    imager is the oracle's with the same question asked through the same SDK calls (fa_probe_segments), not the unoccluded
    one.  LENTIL_OCCLUSION_PROBES=0 renders the unoccluded frame."""
    W, H, M, S = 64, 48, 9, 48
    monkeypatch.setenv("LENTIL_SAMPLES_OVERRIDE", str(S))
    p, model, table, keep = common.po_setup(W, H, samples_override=S, focal_length=np.float32(35.0))
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.03)
    n = W * H * M
    pix = np.arange(n) // M
    px = (pix % W).astype(np.int32); py = (pix // W).astype(np.int32)
    rng = np.random.default_rng(11)
    ox = rng.uniform(-0.5, 0.5, n).astype(np.float32); oy = rng.uniform(-0.5, 0.5, n).astype(np.float32)
    invd = np.full(n, 1.0 / 9.0, np.float32)
    zeros = np.zeros((n, 4), np.float32)
    z4 = np.repeat(cols["pos_z"][:, 3:4], 4, axis=1).copy()
    aov = {"RGBA": (AI_TYPE["RGBA"], cols["rgba"]), "P": (AI_TYPE["VECTOR"], cols["pos_z"]), "Z": (AI_TYPE["FLOAT"], z4),
           "lentil_raydir": (AI_TYPE["RGB"], cols["raydir_time"]), "lentil_time": (AI_TYPE["FLOAT"], zeros),
           "volume": (AI_TYPE["RGB"], zeros), "transmission": (AI_TYPE["RGBA"], zeros), "lentil_ignore": (AI_TYPE["FLOAT"], zeros)}
    keep_arrays = [np.ascontiguousarray(a, np.float32) for _, a in aov.values()]
    fa.fa_set_sphere_occluder.argtypes = [C.c_float] * 4
    fa.fa_set_sphere_occluder.restype = None
    fa.fa_probe_counts.argtypes = [C.c_void_p, C.c_int]
    fa.fa_probe_counts.restype = None
    counts = (C.c_uint64 * 2)()

    def render(sphere):
        fa.fa_messages_clear()
        fa.fa_set_sphere_occluder(*[float(v) for v in sphere])
        fa.fa_probe_counts(counts, 1)
        u, cam = _scene(fa, W, H, ["RGBA RGBA gaussian_filter driver_exr"])
        fa.fa_node_set_int(C.c_void_p(cam), b"camera_type", 1)
        fa.fa_node_set_int(C.c_void_p(cam), b"lens_model", 0)
        fa.fa_set_samples(C.c_void_p(u), n, px.ctypes.data_as(C.c_void_p), py.ctypes.data_as(C.c_void_p), ox.ctypes.data_as(C.c_void_p),
                          oy.ctypes.data_as(C.c_void_p), invd.ctypes.data_as(C.c_void_p))
        for (name, (t, _)), a in zip(aov.items(), keep_arrays):
            fa.fa_set_aov(C.c_void_p(u), name.encode(), t, a.ctypes.data_as(C.c_void_p))
        rc = fa.fa_render(C.c_void_p(u), 4, 16)
        assert rc == 0 and fa.fa_error_count() == 0, _messages(fa)
        img = np.zeros((H, W, 4), np.float32)
        assert fa.fa_get_image(C.c_void_p(u), b"RGBA", img.ctypes.data_as(C.c_void_p)) == 0
        fa.fa_universe_destroy(C.c_void_p(u))
        fa.fa_probe_counts(counts, 0)
        return img, int(counts[0]), int(counts[1])

    def oracle(probe):
        # AOVs in the plugin's order: RGBA, lentil_debug (own z-buffer, no column), lentil_raydir (RGB widened with alpha 1)
        kinds = [_abi.FILTER_GAUSSIAN, _abi.FILTER_CLOSEST_DEBUG, _abi.FILTER_GAUSSIAN]
        ocols = dict(cols)
        widen = lambda a: np.ascontiguousarray(np.concatenate([a[:, :3], np.ones((n, 1), np.float32)], 1), np.float32)
        ocols["extra"] = [np.zeros_like(cols["rgba"]), widen(cols["raydir_time"])]
        ovisits, okeep = capi.make_visits(ocols, visits_per_pixel=M, pixels_per_row=W)
        lens = orc.orc_lens_create(C.byref(table))
        ref = oracle_lib.Frame(orc, p, n_aovs=3, kinds=kinds, keep_log=True)
        if probe:
            ref.set_probe(C.cast(fa.fa_probe_segments, C.c_void_p).value, None)
        ref.run(lens, None, ovisits)
        orc.orc_lens_destroy(lens)
        want = ref.resolve(0).reshape(p.yres, p.xres, 4)[:H, :W].copy()
        c = ref.counters()
        ref.close()
        return want, (int(c.attempted_draws), int(c.accepted_draws))

    def same(got, want):
        m = want != 0
        return np.array_equal(got != 0, m) and float(np.max(np.abs(got[m].astype(np.float64) - want[m]) / np.abs(want[m]))) < 1e-5

    sphere = (6.0, 2.0, -70.0, 9.0)          # beside the axis, between the lens and the far highlights (cm, camera at the origin)
    try:
        got, probed, hits = render(sphere)
        assert probed > 1000 and 0 < hits < probed, (probed, hits)
        fa.fa_set_sphere_occluder(*[float(v) for v in sphere])
        want, wc = oracle(True)
        free, fc = oracle(False)
        assert wc != fc                        # the occluder bites ...
        assert same(got, want) and not same(got, free)
        # ... a scene without one probes and finds nothing: the unoccluded frame
        got0, probed0, hits0 = render((0.0, 0.0, 0.0, 0.0))
        assert probed0 > 1000 and hits0 == 0 and same(got0, free)
        # ... and probing switched off asks nothing, occluder or not
        monkeypatch.setenv("LENTIL_OCCLUSION_PROBES", "0")
        got1, probed1, _ = render(sphere)
        assert probed1 == 0 and same(got1, free)
    finally:
        fa.fa_set_sphere_occluder(0.0, 0.0, 0.0, 0.0)
