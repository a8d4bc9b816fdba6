"""Start/end of every kernel of the LAST pass in a rocprofv3 --kernel-trace CSV, relative to the pass's first scan
launch (microseconds), with the HIP queue each ran on.  usage: timeline.py <rocprof output dir>"""
import csv
import glob
import os
import sys


def main():
    files = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
    f = max(files, key=os.path.getmtime)          # (a directory reused by several runs holds several traces)
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    mine = ("scan_", "solve_", "accept_", "resolve_", "prep_", "publish_", "clear_", "reset_round", "closest_", "fold_", "crypto_")
    # the last pass: from its first scan launch (the scan after the last resolve / clear before it) to the end
    scans = [i for i, r in enumerate(rows) if "scan_" in r["Kernel_Name"]]
    if not scans:
        raise SystemExit("no scan kernel in " + f)
    s = scans[-1]
    while s > 0 and "scan_" in rows[s - 1]["Kernel_Name"]:      # (chunked form: several scan launches back to back)
        s -= 1
    t0 = int(rows[s]["Start_Timestamp"])
    for r in rows[s:]:
        name = r["Kernel_Name"]
        if not any(k in name for k in mine) and "rocclr" not in name:
            continue
        a = (int(r["Start_Timestamp"]) - t0) / 1e3
        b = (int(r["End_Timestamp"]) - t0) / 1e3
        print("%9.1f %9.1f %8.1f  q%-2s grid %-8s %s" % (a, b, b - a, r.get("Queue_Id", "?"), r.get("Grid_Size", "?"), name[:60]))


if __name__ == "__main__":
    main()
