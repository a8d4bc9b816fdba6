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
    # a pass starts at a scan kernel that follows a resolve kernel (or the beginning)
    starts = [i for i, r in enumerate(rows) if "scan_" in r["Kernel_Name"] and
              (i == 0 or "scan_" not in rows[i - 1]["Kernel_Name"]) and
              not any("scan_" in rows[j]["Kernel_Name"] or "solve" in rows[j]["Kernel_Name"] or "accept" in rows[j]["Kernel_Name"]
                      for j in range(max(0, i - 3), i))]
    which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
    s = starts[which]
    e = len(rows) if which == -1 or which + 1 >= len(starts) else starts[which + 1]
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
