#!/usr/bin/env python3
"""pota_amd/plugin/lentil.mtd -- the Arnold metadata file shipped next to lentil.so: one [node] block per plugin node,
one [attr] block per parameter of the camera node in declaration order (lentil_camera_node_parameters()), with the
facts a DCC translator needs (labels, enum-ness, linkability, the Maya node ids lentil is registered under).  The
reference generates the same kind of file from src/lentil_camera.ui (src/uigen.py) plus src/lentil_hardcode.mtd;
tests/test_plugin.py checks this one against that (tests/golden/lentil.mtd).
Usage: python tools/gen_mtd.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pota_amd import bridge  # noqa: E402

TYPE_NAMES = {0x01: "INT", 0x03: "BOOL", 0x04: "FLOAT", 0x0A: "STRING", 0x0F: "ENUM"}
GROUPS = [("General", ["camera_type", "bidir_sample_mult", "units", "sensor_width", "enable_dof", "fstop", "focus_dist",
                       "aperture_blades_lentil", "exp"]),
          ("Polynomial Optics", ["lens_model", "wavelength", "extra_sensor_shift"]),
          ("Thin Lens", ["focal_length_lentil", "optical_vignetting", "abb_spherical", "abb_distortion", "abb_coma",
                         "bokeh_circle_to_square", "bokeh_anamorphic"]),
          ("Bokeh Texture", ["bokeh_enable_image", "bokeh_image_path"]),
          ("Bidirectional", ["vignetting_retries", "abb_chromatic", "abb_chromatic_type", "bidir_add_energy",
                             "bidir_add_energy_minimum_luminance", "bidir_add_energy_transition", "enable_bidir_transmission",
                             "enable_skydome"])]


def main():
    params = bridge.camera_node_parameters()
    names = [p["name"] for p in params]
    assert sorted(n for _, g in GROUPS for n in g) == sorted(names), "GROUPS out of date"
    L = ["[node lentil_camera]", '\tdesc STRING "Lentil camera (MI355X redistribution path)"', '\tmaya.name STRING "camera"',
         '\tmaya.classification STRING "camera"', '\tmaya.translator STRING "lentil_camera"', '\tmaya.attr_prefix STRING ""',
         "\tmaya.id INT 0x00070507", '\thoudini.category STRING "Lentil"']
    for k, (g, members) in enumerate(GROUPS):
        L.append('\thoudini.parm.group.g%d STRING "%s;%d"' % (k, g, len(members)))
    L.append('\thoudini.order STRING "' + " ".join("g%d %s" % (k, " ".join(m)) for k, (_, m) in enumerate(GROUPS)) + '"')
    L.append("")
    for p in params:
        L.append("\t[attr %s]" % p["name"])
        L.append('\t\thoudini.label STRING "%s"' % p["name"].replace("_", " ").title())
        L.append('\t\tdesc STRING "%s parameter of lentil_camera (%s)"' % (p["name"], TYPE_NAMES.get(p["type"], "?")))
        L.append("\t\tlinkable BOOL FALSE")
        L.append("")
    L += ["[node imager_lentil]", '\tmaya.name STRING "imager_lentil"', '\tmaya.classification STRING "imager"',
          '\tmaya.attr_prefix STRING ""', '\tmaya.output_name STRING "out"', '\tmaya.output_shortname STRING "out"',
          "\tmaya.id INT 0x00070512", "", "\t[attr enable]", "\t\tlinkable BOOL FALSE", "",
          "[node lentil_filter]", '\tmaya.attr_prefix STRING ""', '\tmaya.translator STRING "lentil_filter"',
          "\tmaya.id INT 0x00070948", "",
          "[node lentil_operator]", '\tmaya.name STRING "lentil_operator"', '\tmaya.classification STRING "operator"',
          '\tmaya.attr_prefix STRING ""', "\tmaya.id INT 0x00070513", ""]
    path = os.path.join(ROOT, "pota_amd", "plugin", "lentil.mtd")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write("\n".join(L))
    print(path, len(L), "lines")


if __name__ == "__main__":
    main()
