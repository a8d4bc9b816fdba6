#!/bin/bash
# The kernels of the LAST pass of tools/crypto_rate.py with one cryptomatte AOV, from rocprofv3 --kernel-trace: start, end, queue
# (us from the pass's clear).  usage: tools/crypto_sequence.sh
export HSA_ENABLE_COREDUMP=0 TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
export LENTIL_CRYPTO_RATE_ONLY=0,1
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp; rm -rf /tmp/cseq
rocprofv3 --kernel-trace --output-format csv -d /tmp/cseq -- python3 $R/tools/crypto_rate.py > /dev/null 2>&1
python3 - "$(find /tmp/cseq -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:48], r.get("Stream_Id", r.get("Queue_Id", "")))
        for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
clears = [i for i, r in enumerate(rows) if "clear_touched" in r[2] or "crypto_clear" in r[2]]
i0 = max(i for i, r in enumerate(rows) if "clear_touched" in r[2])
t0 = rows[i0][0]
print("%-50s %9s %9s %8s  queue" % ("kernel", "start us", "end us", "dur us"))
for s, e, n, q in rows[i0:]:
    print("%-50s %9.1f %9.1f %8.1f  %s" % (n, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q))
PY
