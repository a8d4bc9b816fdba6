#!/usr/bin/env python3
"""Condense rocprofv3 output directories into the small files kept under profiles/.

usage: profile_summary.py <round-tag> <stats_dir> [<pmc_fetch_dir> <pmc_write_dir>] [--workload TAG]
  stats_dir : rocprofv3 --kernel-trace --stats --output-format csv -d <stats_dir> -- python3 bench.py ...
  pmc dirs  : rocprofv3 --pmc FETCH_SIZE (resp. WRITE_SIZE) --output-format csv -d <dir> -- python3 bench.py ...
Writes profiles/<tag>_kernel_stats.csv (this library's kernels only), and -- with the pmc dirs --
profiles/<tag>_pmc_scan.json + profiles/pmc_scan_latest.json (HBM bytes per scan launch; FETCH_SIZE is
doubled on gfx950 as MI355X_MICROARCH.md section "HBM" prescribes for 16 B/lane streaming reads).
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
MINE = ("scan_", "solve_", "accept_", "prep_items", "resolve_kernel", "resolve_touched", "reset_round", "crypto_", "publish_kernel", "fold_direct", "clear_touched",
        "pack_rows", "merge_", "compact_rows", "closest_gather", "debug_gather")


def find(d, pat):
    r = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return r[0] if r else None


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    workload = None
    if "--workload" in sys.argv:
        workload = sys.argv[sys.argv.index("--workload") + 1]
        args = [a for a in args if a != workload]
    tag, stats_dir = args[0], args[1]
    out_dir = os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    st = find(stats_dir, "*kernel_stats.csv")
    rows = [r for r in csv.DictReader(open(st)) if any(k in r["Name"] for k in MINE)]
    with open(os.path.join(out_dir, tag + "_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in rows:
            r["Name"] = r["Name"].split("(")[0][:90]
            w.writerow(r)
    print("wrote", tag + "_kernel_stats.csv")
    for r in rows:
        print("  %-70s calls %5s avg %10.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
    if len(args) >= 4:
        def per_launch(d, counter):
            p = find(d, "*counter_collection.csv")
            vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(p))
                    if r["Counter_Name"] == counter and "scan_" in r["Kernel_Name"]]
            return sum(vals) / len(vals), len(vals)
        fetch_kb, nf = per_launch(args[2], "FETCH_SIZE")
        write_kb, nw = per_launch(args[3], "WRITE_SIZE")
        res = {
            "kernel": (workload or "scan_").split()[0], "workload": workload, "launches": [nf, nw],
            "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB_raw": write_kb,
            "fetch_bytes_corrected": fetch_kb * 1024 * 2, "write_bytes": write_kb * 1024,
            "hbm_bytes_per_launch": fetch_kb * 1024 * 2 + write_kb * 1024,
            "note": "FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B for wide coalesced reads), WRITE_SIZE as is",
        }
        for name in (tag + "_pmc_scan.json", "pmc_scan_latest.json"):
            with open(os.path.join(out_dir, name), "w") as f:
                json.dump(res, f, indent=1)
        print(json.dumps(res))


if __name__ == "__main__":
    main()
