#!/usr/bin/env python3
"""Extract known-answer vectors from the reference's Newton-solver trace.

Input : /root/reference/tests/aperture_sampling_debug/writout.txt (a hand-captured log of the
        generated lt_sample_aperture loop; numbers only, 6 decimals).
Output: tests/golden/writout_kats.json -- per logged iteration the printed intermediates, so the
        oracle's 2x2 inverse, Newton update steps (incl. the 0.72 damping), sphereToCs, csToSphere,
        normalise and error-flag rules can be checked without the (absent) polynomial coefficients.
Run in the build container only (the reference tree does not travel to the GPU box).
"""
import json
import os
import re
import sys

SRC = "/root/reference/tests/aperture_sampling_debug/writout.txt"
DST = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "writout_kats.json")

num = r"(-?\d+\.\d+)"
KEYS2 = ["pred_ap", "delta_ap", "out_new_pos", "out_new_dir", "delta_out"]
KEYS3 = ["pred_out_cs_pos", "pred_out_cs_dir", "view", "view_normalized"]
KEYS4 = ["dx1_domaga0", "invApJ", "out", "domega2_dx0", "invJ"]
KEYS1 = ["begin_x", "begin_y", "begin_dx", "begin_dy", "begin_lambda", "sqr_ap_err", "invdetap", "dx", "dy",
         "sqr_err", "invdet", "x", "y"]


def main():
    lines = open(SRC).read().splitlines()
    header = {"aperture_rad": float(re.search(num, lines[0]).group(1)),
              "sensorshift": float(re.search(num, lines[3]).group(1))}
    cases, cur, it = [], None, None
    for ln in lines:
        if re.match(r"^(working|not working|completely healthy)", ln):
            cur = {"label": ln.strip().rstrip(":"), "iterations": []}
            cases.append(cur)
            it = None
            continue
        m = re.search(r"\|\s+(.*)$", ln)
        if not m or cur is None:
            mm = re.match(r"cam space pos: (\S+) (\S+) (\S+)", ln)
            if mm and cur is not None:
                cur["cam_space_pos"] = [float(mm.group(i)) for i in (1, 2, 3)]
            continue
        body = m.group(1).strip()
        if body.startswith("count (k):"):
            it = {"k": int(body.split(":")[1]), "errors": [], "reset": False}
            cur["iterations"].append(it)
            continue
        if body.startswith("error |="):
            it["errors"].append(int(body.split("=")[1]))
            continue
        if body.startswith("error reset"):
            it["reset"] = True
            continue
        key, _, rest = body.partition(":")
        vals = [float(v) for v in re.findall(num, rest)]
        if it is None:
            if key in ("view", "view_normalized"):
                cur.setdefault("pre_" + key, vals)
            continue
        if key == "view" and "view" in it:      # (not expected)
            continue
        it[key] = vals if len(vals) != 1 else vals[0]
    out = {"source": "tests/aperture_sampling_debug/writout.txt", "header": header, "cases": cases}
    with open(DST, "w") as f:
        json.dump(out, f, indent=0)
    print("cases:", [(c["label"], len(c["iterations"])) for c in cases], "->", os.path.normpath(DST))


if __name__ == "__main__":
    sys.exit(main())
