#!/usr/bin/env python3
"""Input fixtures from the two captured data files the reference's tests hold:

  tests/cuda/sampledata.txt                                  52 967 captured AOV samples of the light-grid scene, one
                                                              per line: r g b a Z Px Py Pz (camera space, cm)
  tests/po_bidir_debug/po_bidir_spheres_debug_position.txt   3 699 camera-space positions, "[x,y,z],"

Both are INPUTS only (the reference holds no expected outputs for them).  Written here, in the build container,
as small arrays under tests/golden/ -- data, not source:
  sampledata_2k.npy        every 26th line of sampledata.txt -> [2038, 8] float32
  po_bidir_positions.npy   all positions -> [3699, 3] float32
Usage: python tools/make_sample_fixtures.py [/root/reference]
"""
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    rows = []
    with open(os.path.join(ref, "tests", "cuda", "sampledata.txt")) as f:
        for i, line in enumerate(f):
            if i % 26:
                continue
            v = line.split()
            if len(v) == 8:
                rows.append([float(x) for x in v])
    a = np.asarray(rows, np.float32)
    np.save(os.path.join(ROOT, "tests", "golden", "sampledata_2k.npy"), a)
    pos = []
    with open(os.path.join(ref, "tests", "po_bidir_debug", "po_bidir_spheres_debug_position.txt")) as f:
        for line in f:
            m = re.findall(r"[-+0-9.eE]+", line)
            if len(m) == 3:
                pos.append([float(x) for x in m])
    b = np.asarray(pos, np.float32)
    np.save(os.path.join(ROOT, "tests", "golden", "po_bidir_positions.npy"), b)
    print("sampledata_2k.npy", a.shape, "po_bidir_positions.npy", b.shape)


if __name__ == "__main__":
    main()
