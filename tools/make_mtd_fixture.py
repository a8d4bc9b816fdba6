#!/usr/bin/env python3
"""tests/golden/lentil.mtd: the reference's own Arnold metadata file, built the way its CMake does
(src/CMakeLists.txt:17-67): `uigen.py` turns src/lentil_camera.ui into lentil_camera.mtd, and lentil_hardcode.mtd is
appended.  Run in the build container (the reference tree is not on the GPU box); the result is a data fixture that
tests/test_plugin.py compares with the plugin's parameter table and with the .mtd this build ships.
Usage: python tools/make_mtd_fixture.py [/root/reference]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    src = os.path.join(ref, "src")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "lentil_camera.mtd")
        subprocess.check_call([sys.executable, os.path.join(src, "uigen.py"), "--ui_file", os.path.join(src, "lentil_camera.ui"),
                               "--mtd_output", out, "--ae_template_output", os.path.join(d, "ae.py"),
                               "--args_output", os.path.join(d, "a.args"), "--c4d_output", os.path.join(d, "c4d")],
                              cwd=d, env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
        text = open(out).read() + open(os.path.join(src, "lentil_hardcode.mtd")).read()
    path = os.path.join(ROOT, "tests", "golden", "lentil.mtd")
    with open(path, "w") as f:
        f.write(text)
    print(path, len(text.splitlines()), "lines")


if __name__ == "__main__":
    main()
