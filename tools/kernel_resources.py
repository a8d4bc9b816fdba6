#!/usr/bin/env python3
"""Registers / spills / LDS / occupancy of every kernel in liblentil_hip.so, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks (compiles to a scratch file, no GPU needed)."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

def main():
    src = os.path.join(ROOT, "pota_amd", "csrc", "lentil_hip.hip")
    g.write_embedded()       # (generated/embedded_sources.inc from the sources as they are)
    with tempfile.TemporaryDirectory() as d:
        cmd = [g.HIPCC] + g.HIP_FLAGS + ["-I", os.path.join(ROOT, "include"), "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:] + [
                                         src, "-o", os.path.join(d, "t.so")]
        err = subprocess.run(cmd, stderr=subprocess.PIPE, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip()
            cur = {"name": name}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[a-z/A-Z]+\])?: (\d+) \[-Rpass", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    print("%-72s %5s %5s %6s %6s %7s %4s" % ("kernel", "VGPR", "AGPR", "SGPR", "spillV", "LDS", "occ"))
    for r in rows:
        n = re.sub(r"\(.*", "", r["name"])[:72]
        print("%-72s %5d %5d %6d %6d %7d %4d" % (n, r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("TotalSGPRs", -1),
                                                r.get("VGPRs Spill", r.get("VGPR Spill", -1)), r.get("LDS Size", -1),
                                                r.get("Occupancy", -1)))

if __name__ == "__main__":
    main()
