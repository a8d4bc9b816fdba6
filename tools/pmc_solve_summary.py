#!/usr/bin/env python3
"""Sum the counters of a rocprofv3 --pmc counter_collection.csv per kernel family (solve_po, solve_slow, accept, scan)."""
import csv, json, sys, collections


def family(name):
    for k in ("solve_po_kernel", "solve_slow_kernel", "accept_kernel", "scan_", "prep_items", "resolve"):
        if k in name:
            if k == "solve_po_kernel":
                lens = "petzval" if "petzval" in name else ("double_gauss" if "double_gauss" in name else "tables")
                return k + "<" + lens + ">"
            return k
    return None


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for r in csv.DictReader(open(sys.argv[1])):
        f = family(r["Kernel_Name"])
        if not f:
            continue
        acc[f][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[f].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
    out = {f: dict(v, launches=len(launches[f])) for f, v in acc.items()}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
