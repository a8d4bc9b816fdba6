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


# Per parameter: what a DCC translator acts on -- label, hard and soft range, linkability, when the widget is greyed out,
# file-browser hints.  The same keys and values as the reference's lentil.mtd carries (tests/golden/lentil.mtd, from
# src/lentil_camera.ui through src/uigen.py; tests/test_plugin.py diffs them); the free-text descriptions are this repo's.
ATTR_META = {'camera_type': [('houdini.label', 'STRING', '"Camera Type"'), ('linkable', 'BOOL', 'FALSE')],
 'bidir_sample_mult': [('houdini.label', 'STRING', '"Samples Multiplier"'),
                       ('min', 'FLOAT', '0'),
                       ('max', 'FLOAT', '300'),
                       ('softmin', 'FLOAT', '0'),
                       ('softmax', 'FLOAT', '10'),
                       ('linkable', 'BOOL', 'FALSE'),
                       ('houdini.disable_when', 'STRING', '"{ enable_dof == 0 }"')],
 'units': [('houdini.label', 'STRING', '"Scene Units"'), ('linkable', 'BOOL', 'FALSE')],
 'sensor_width': [('houdini.label', 'STRING', '"Sensor Width (mm)"'),
                  ('min', 'FLOAT', '0.1'),
                  ('max', 'FLOAT', '100'),
                  ('softmin', 'FLOAT', '16'),
                  ('softmax', 'FLOAT', '70'),
                  ('linkable', 'BOOL', 'TRUE')],
 'enable_dof': [('houdini.label', 'STRING', '"Enable Depth of Field"'), ('linkable', 'BOOL', 'FALSE')],
 'fstop': [('houdini.label', 'STRING', '"F-Stop"'),
           ('min', 'FLOAT', '0'),
           ('max', 'FLOAT', '64'),
           ('softmin', 'FLOAT', '0.7'),
           ('softmax', 'FLOAT', '32'),
           ('linkable', 'BOOL', 'TRUE'),
           ('houdini.disable_when', 'STRING', '"{ enable_dof == 0 }"')],
 'focus_dist': [('houdini.label', 'STRING', '"Focus Distance (cm)"'),
                ('min', 'FLOAT', '0'),
                ('max', 'FLOAT', '100000'),
                ('softmin', 'FLOAT', '0'),
                ('softmax', 'FLOAT', '2000'),
                ('linkable', 'BOOL', 'TRUE'),
                ('houdini.disable_when', 'STRING', '"{ enable_dof == 0 }"')],
 'aperture_blades_lentil': [('houdini.label', 'STRING', '"Aperture Blades"'),
                            ('min', 'FLOAT', '0'),
                            ('max', 'FLOAT', '50'),
                            ('softmin', 'FLOAT', '0'),
                            ('softmax', 'FLOAT', '8'),
                            ('linkable', 'BOOL', 'FALSE'),
                            ('houdini.disable_when', 'STRING', '"{ enable_dof == 0 }"')],
 'exp': [('houdini.label', 'STRING', '"Exposure"'),
         ('min', 'FLOAT', '0'),
         ('max', 'FLOAT', '99999'),
         ('softmin', 'FLOAT', '0'),
         ('softmax', 'FLOAT', '5'),
         ('linkable', 'BOOL', 'TRUE')],
 'lens_model': [('houdini.label', 'STRING', '"Lens Model"'),
                ('linkable', 'BOOL', 'FALSE'),
                ('houdini.disable_when', 'STRING', '"{ cameratype == ThinLens }"')],
 'wavelength': [('houdini.label', 'STRING', '"Wavelength (nm)"'),
                ('min', 'FLOAT', '390'),
                ('max', 'FLOAT', '700'),
                ('linkable', 'BOOL', 'TRUE'),
                ('houdini.disable_when', 'STRING', '"{ cameratype == ThinLens }"')],
 'extra_sensor_shift': [('houdini.label', 'STRING', '"Additional Sensor shift (mm)"'),
                        ('min', 'FLOAT', '-10'),
                        ('max', 'FLOAT', '10'),
                        ('softmin', 'FLOAT', '-3'),
                        ('softmax', 'FLOAT', '3'),
                        ('linkable', 'BOOL', 'TRUE'),
                        ('houdini.disable_when', 'STRING', '"{ cameratype == ThinLens }"')],
 'focal_length_lentil': [('houdini.label', 'STRING', '"Focal Length (mm)"'),
                         ('min', 'FLOAT', '0.01'),
                         ('max', 'FLOAT', '99999'),
                         ('softmin', 'FLOAT', '5'),
                         ('softmax', 'FLOAT', '500'),
                         ('linkable', 'BOOL', 'TRUE'),
                         ('houdini.disable_when', 'STRING', '"{ cameratype == PolynomialOptics }"')],
 'optical_vignetting': [('houdini.label', 'STRING', '"Optical Vignetting"'),
                        ('min', 'FLOAT', '0'),
                        ('max', 'FLOAT', '20'),
                        ('softmin', 'FLOAT', '0'),
                        ('softmax', 'FLOAT', '5'),
                        ('linkable', 'BOOL', 'TRUE'),
                        ('houdini.disable_when', 'STRING', '"{ cameratype == PolynomialOptics }{ enable_dof == 0 }"')],
 'abb_spherical': [('houdini.label', 'STRING', '"Aberration (spherical)"'),
                   ('min', 'FLOAT', '0'),
                   ('max', 'FLOAT', '1'),
                   ('softmin', 'FLOAT', '0'),
                   ('softmax', 'FLOAT', '1'),
                   ('linkable', 'BOOL', 'TRUE'),
                   ('houdini.disable_when', 'STRING', '"{ cameratype == PolynomialOptics }{ enable_dof == 0 }"')],
 'abb_distortion': [('houdini.label', 'STRING', '"Aberration (distortion)"'),
                    ('min', 'FLOAT', '-50'),
                    ('max', 'FLOAT', '50'),
                    ('softmin', 'FLOAT', '-10'),
                    ('softmax', 'FLOAT', '10'),
                    ('linkable', 'BOOL', 'TRUE'),
                    ('houdini.disable_when', 'STRING', '"{ cameratype == PolynomialOptics }{ enable_dof == 0 }"')],
 'abb_coma': [('houdini.label', 'STRING', '"Aberration (coma)"'),
              ('min', 'FLOAT', '-1'),
              ('max', 'FLOAT', '1'),
              ('linkable', 'BOOL', 'TRUE'),
              ('houdini.disable_when', 'STRING', '"{ cameratype == PolynomialOptics }{ enable_dof == 0 }"')],
 'bokeh_circle_to_square': [('houdini.label', 'STRING', '"Square Bokeh"'),
                            ('min', 'FLOAT', '0'),
                            ('max', 'FLOAT', '1'),
                            ('softmin', 'FLOAT', '0'),
                            ('softmax', 'FLOAT', '1'),
                            ('linkable', 'BOOL', 'TRUE'),
                            ('houdini.disable_when', 'STRING', '"{ cameratype == PolynomialOptics }{ enable_dof == 0 }"')],
 'bokeh_anamorphic': [('houdini.label', 'STRING', '"Anamorphic Squeeze"'),
                      ('min', 'FLOAT', '0'),
                      ('max', 'FLOAT', '1'),
                      ('softmin', 'FLOAT', '0'),
                      ('softmax', 'FLOAT', '1'),
                      ('linkable', 'BOOL', 'TRUE'),
                      ('houdini.disable_when', 'STRING', '"{ cameratype == PolynomialOptics }{ enable_dof == 0 }"')],
 'bokeh_enable_image': [('houdini.label', 'STRING', '"Enable Texture"'), ('linkable', 'BOOL', 'FALSE'), ('houdini.join_next', 'BOOL', 'TRUE')],
 'bokeh_image_path': [('houdini.label', 'STRING', '"Filepath"'),
                      ('linkable', 'BOOL', 'FALSE'),
                      ('houdini.disable_when', 'STRING', '"{ bokeh_enable_image == 0 }{ enable_dof == 0 }"'),
                      ('c4d.gui_type', 'INT', '3'),
                      ('c4d.label', 'STRING', '"File Path"'),
                      ('maya.usedAsFilename', 'BOOL', 'TRUE'),
                      ('c4d.gui_type', 'INT', '3'),
                      ('houdini.type', 'STRING', '"file:image"'),
                      ('houdini.callback', 'STRING', '"python:import htoa.texture; htoa.texture.imageFilenameCallback()"')],
 'vignetting_retries': [('houdini.label', 'STRING', '"Vignetting Quality"'),
                        ('min', 'FLOAT', '1'),
                        ('max', 'FLOAT', '500'),
                        ('softmin', 'FLOAT', '1'),
                        ('softmax', 'FLOAT', '50'),
                        ('linkable', 'BOOL', 'FALSE'),
                        ('houdini.disable_when', 'STRING', '"{ bidir_sample_mult == 0 }{ enable_dof == 0 }"')],
 'abb_chromatic': [('houdini.label', 'STRING', '"Aberration (chromatic)"'),
                   ('min', 'FLOAT', '0'),
                   ('max', 'FLOAT', '3'),
                   ('softmin', 'FLOAT', '0'),
                   ('softmax', 'FLOAT', '1'),
                   ('linkable', 'BOOL', 'TRUE'),
                   ('houdini.disable_when', 'STRING', '"{ cameratype == PolynomialOptics }{ enable_dof == 0 }{ bidir_sample_mult == 0 }"')],
 'abb_chromatic_type': [('houdini.label', 'STRING', '"Chromatic shift"'), ('linkable', 'BOOL', 'FALSE')],
 'bidir_add_energy': [('houdini.label', 'STRING', '"Additional Energy"'),
                      ('min', 'FLOAT', '0'),
                      ('max', 'FLOAT', '100'),
                      ('softmin', 'FLOAT', '0'),
                      ('softmax', 'FLOAT', '10'),
                      ('linkable', 'BOOL', 'TRUE'),
                      ('houdini.disable_when', 'STRING', '"{ bidir_sample_mult == 0 }{ enable_dof == 0 }"')],
 'bidir_add_energy_minimum_luminance': [('houdini.label', 'STRING', '"Additional Energy Treshold"'),
                                        ('min', 'FLOAT', '0'),
                                        ('max', 'FLOAT', '100'),
                                        ('softmin', 'FLOAT', '0'),
                                        ('softmax', 'FLOAT', '5'),
                                        ('linkable', 'BOOL', 'TRUE'),
                                        ('houdini.disable_when', 'STRING', '"{ bidir_sample_mult == 0 }{ enable_dof == 0 }"')],
 'bidir_add_energy_transition': [('houdini.label', 'STRING', '"Additional Energy Treshold Transition"'),
                                 ('min', 'FLOAT', '0'),
                                 ('max', 'FLOAT', '10'),
                                 ('softmin', 'FLOAT', '0'),
                                 ('softmax', 'FLOAT', '5'),
                                 ('linkable', 'BOOL', 'TRUE'),
                                 ('houdini.disable_when', 'STRING', '"{ bidir_sample_mult == 0 }{ enable_dof == 0 }"')],
 'enable_bidir_transmission': [('houdini.label', 'STRING', '"Enable for transmitted surfaces"'),
                               ('linkable', 'BOOL', 'FALSE'),
                               ('houdini.disable_when', 'STRING', '"{ bidir_sample_mult == 0 }{ enable_dof == 0 }"')],
 'enable_skydome': [('houdini.label', 'STRING', '"Enable Skydome Redistribution"'), ('linkable', 'BOOL', 'FALSE')]}


def main():
    params = bridge.camera_node_parameters()
    names = [p["name"] for p in params]
    assert sorted(n for _, g in GROUPS for n in g) == sorted(names), "GROUPS out of date"
    assert sorted(ATTR_META) == sorted(names), "ATTR_META out of date"
    L = ["[node lentil_camera]", '\tdesc STRING "Lentil camera: thin lens or polynomial optics, bidirectional bokeh redistribution on the GPU"',
         '\tc4d.classification STRING "generic"', '\tmaya.name STRING "camera"',
         '\tmaya.classification STRING "camera"', '\tmaya.translator STRING "lentil_camera"', '\tmaya.attr_prefix STRING ""',
         "\tmaya.id INT 0x00070507", '\thoudini.icon STRING "SHOP_lens"', '\thoudini.category STRING "Lentil"',
         '\thoudini.help_url STRING "http://www.lentil.xyz"']
    for k, (g, members) in enumerate(GROUPS):
        L.append('\thoudini.parm.group.g%d STRING "%s;%d"' % (k, g, len(members)))
    L.append('\thoudini.order STRING ' + "\n\t".join('"g%d %s "' % (k, " ".join(m)) for k, (_, m) in enumerate(GROUPS)))
    L.append("")
    for p in params:
        L.append("\t[attr %s]" % p["name"])
        meta = ATTR_META[p["name"]]
        text = "%s (%s, default %s)" % (dict((k, v) for k, _, v in meta)["houdini.label"].strip('"'), TYPE_NAMES.get(p["type"], "?"),
                                        p["default_string"] if p["type"] == 0x0A else ("%g" % p["default"]))
        for key, typ, val in meta:
            L.append("\t\t%s %s %s" % (key, typ, val))
        # the free text (ours): after the ranges, where the reference's file has it
        at = len(L) - sum(1 for key, _, _ in meta if key in ("linkable", "houdini.disable_when", "houdini.join_next", "c4d.gui_type", "c4d.label",
                                                              "maya.usedAsFilename", "houdini.type", "houdini.callback"))
        L[at:at] = ['\t\tdesc STRING "%s"' % text, '\t\thoudini.help STRING "%s"' % text]
        L.append("")
    # imager, filter: src/lentil_hardcode.mtd's facts (node ids, translator names, hidden attributes); the operator's block
    # is commented out there, and so it is here (MtoA then treats lentil_operator as a plain operator node)
    L += ["[node imager_lentil]", '\tmaya.name STRING "imager_lentil"', '\tmaya.classification STRING "imager"',
          '\tmaya.attr_prefix STRING ""', '\tmaya.output_name STRING "out"', '\tmaya.output_shortname STRING "out"',
          "\tmaya.id INT 0x00070512", "", "\t[attr layer_selection]", "\t\tmaya.hide BOOL false", "\t[attr input]",
          "\t\tmaya.hide BOOL TRUE", "",
          "[node lentil_filter]", '\tmaya.attr_prefix STRING ""', '\tmaya.translator STRING "lentil_filter"',
          "\tmaya.id INT 0x00070948", "",
          "#[node lentil_operator]", '#\tmaya.name STRING "lentil_operator"', '#\tmaya.attr_prefix STRING ""',
          '#\tmaya.classification STRING "operator"', "#\tmaya.id INT 0x00070513", ""]
    path = os.path.join(ROOT, "pota_amd", "plugin", "lentil.mtd")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write("\n".join(L))
    print(path, len(L), "lines")


if __name__ == "__main__":
    main()
