#!/bin/bash
# per-launch durations of the cryptomatte replay kernels (rocprofv3 --kernel-trace over tools/crypto_rate.py), both own-pixel kernels
# (the runtime reads it when the profiler's preloaded library initialises it -- before the program's own os.environ.setdefault runs)
export GPU_MAX_HW_QUEUES=8
export HSA_ENABLE_COREDUMP=0 TMPDIR=/tmp
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp
for e in 1 0; do
  export LENTIL_CRYPTO_REG=$e
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cr_$e -- python3 $R/tools/crypto_rate.py > /dev/null 2>&1
  f=$(find /tmp/cr_$e -name "*kernel_trace.csv" | head -1)
  echo "LENTIL_CRYPTO_REG=$e (us per launch)"
  python3 - "$f" <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].split("(")[0]
    if "crypto" in n or "flag_bits" in n:
        d[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    print("  %-45s" % k[:45], " ".join("%.0f" % x for x in v))
PY
done
