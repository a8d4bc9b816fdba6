#!/bin/bash
# PMC passes over the cryptomatte replay kernels (tools/crypto_rate.py: headline frame, 1 and 3 cryptomatte AOVs)
cd "$(dirname "$0")/.."
# (the runtime reads it when the profiler's preloaded library initialises it -- before the program's own os.environ.setdefault runs)
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp LENTIL_CRYPTO_OVERLAP=0
O=gpurun_out/pmc_crypto; mkdir -p $O
SET1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT"
SET2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
i=1
for S in "$SET1" "$SET2" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmc_cr_$i
  timeout 600 rocprofv3 --pmc $S --output-format csv -d /tmp/pmc_cr_$i -- python3 tools/crypto_rate.py > $O/set$i.log 2>&1
  f=$(find /tmp/pmc_cr_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" > $O/set$i.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:60]
    if "crypto" not in k and "flag_bits" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(k, r["Counter_Name"])] += 1
for k, v in acc.items():
    print(k)
    for c, x in sorted(v.items()):
        print("   %-24s %16.0f  (%d dispatches)" % (c, x, n[(k, c)]))
PY
  i=$((i+1))
done
cat $O/set*.txt
