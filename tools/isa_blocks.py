#!/usr/bin/env python3
"""Per-basic-block instruction counts of one kernel in a device assembly dump (hipcc -S --offload-device-only).
usage: isa_blocks.py dev.s <substring of the mangled kernel name> [min instructions]"""
import re, sys
path, key = sys.argv[1], sys.argv[2]
mn = int(sys.argv[3]) if len(sys.argv) > 3 else 10
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and l.split(":")[0].endswith(key.split("*")[-1]) or (l.startswith("_Z") and re.search(key, l.split(":")[0] or "")))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
lbl, st, cnt = "entry", start, {}
def out():
    n = sum(cnt.values())
    if n >= mn:
        print("%-14s line %6d  n=%5d valu=%5d f64=%5d salu=%4d smem=%3d lds=%3d vmem=%3d" % (lbl, st - start, n, cnt.get("v", 0), cnt.get("f64", 0) , cnt.get("s", 0), cnt.get("smem", 0), cnt.get("ds", 0), cnt.get("vm", 0)))
tot = {}
for i in range(start + 1, end + 1):
    l = lines[i].strip()
    if re.match(r"^\.LBB\d+_\d+:", l):
        out(); lbl, st, cnt = l.split(":")[0], i, {}
        continue
    if not l or l.startswith(";") or l.startswith("."):
        continue
    op = l.split()[0]
    if op.startswith("v_"):
        cnt["v"] = cnt.get("v", 0) + 1
        if "f64" in op: cnt["f64"] = cnt.get("f64", 0) + 1
    elif op.startswith("s_load") or op.startswith("s_buffer"): cnt["smem"] = cnt.get("smem", 0) + 1
    elif op.startswith("s_"): cnt["s"] = cnt.get("s", 0) + 1
    elif op.startswith("ds_"): cnt["ds"] = cnt.get("ds", 0) + 1
    else: cnt["vm"] = cnt.get("vm", 0) + 1
out()
print("kernel lines", end - start)
