#!/bin/bash
# The kernels of the LAST streamed pass of a short bench run, from rocprofv3 --kernel-trace of the production library: start, end
# and the gap to the previous kernel's end (us, relative to the pass's first kernel).  usage: tools/pass_sequence.sh [bench flags]
# (the runtime reads it when the profiler's preloaded library initialises it -- before the program's own os.environ.setdefault runs)
export GPU_MAX_HW_QUEUES=8
export HSA_ENABLE_COREDUMP=0 TMPDIR=/tmp
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp; rm -rf /tmp/pseq
rocprofv3 --kernel-trace --output-format csv -d /tmp/pseq -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-second-regime --no-configs --no-pcie --no-parity-check --no-scan-alone "$@" > /dev/null 2>&1
python3 - "$(find /tmp/pseq -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:48], r.get("Stream_Id", r.get("Queue_Id", "")))
        for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# the last PASSES passes (default 1): each from its clear_touched_kernel on
import os
clears = [i for i, r in enumerate(rows) if "clear_touched" in r[2]]
n_pass = int(os.environ.get("PASSES", "1"))
for k, i0 in enumerate(clears[-n_pass:]):
    i1 = clears[clears.index(i0) + 1] if clears.index(i0) + 1 < len(clears) else len(rows)
    t0 = rows[i0][0]
    print("%-50s %9s %9s %8s  queue" % ("kernel", "start us", "end us", "dur us"))
    for s, e, n, q in rows[i0:i1]:
        print("%-50s %9.1f %9.1f %8.1f  %s" % (n, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q))
PY
