#!/usr/bin/env python3
"""Print the kernel timeline of the last redistribution pass from a rocprofv3 --kernel-trace csv.

usage: timeline.py <dir with *_kernel_trace.csv>
Times are microseconds relative to the first scan launch of the pass.
"""
import csv
import glob
import os
import sys


def main():
    f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # a pass ends with its resolve kernel: pass `which` (default: the last one) = everything after the previous resolve,
    # starting at its first scan launch
    ends = [i for i, r in enumerate(rows) if "resolve_kernel" in r["Kernel_Name"]]
    which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
    e = ends[which] + 1
    k = ends.index(ends[which])
    lo = ends[k - 1] + 1 if k > 0 else 0
    s = next(i for i in range(lo, e) if "scan_" in rows[i]["Kernel_Name"])
    t0 = int(rows[s]["Start_Timestamp"])
    for r in rows[s:e]:
        a = (int(r["Start_Timestamp"]) - t0) / 1e3
        b = (int(r["End_Timestamp"]) - t0) / 1e3
        name = r["Kernel_Name"].split("(")[0][:60]
        print("%9.1f %9.1f %8.1f  q%-3s grid %-8s %s" % (a, b, b - a, r.get("Queue_Id", "?"), r.get("Grid_Size", "?"), name))
        if "resolve" in name:
            break


if __name__ == "__main__":
    main()
