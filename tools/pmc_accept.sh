#!/bin/bash
# PMC passes over the accept kernel of the headline frame in the chunked form (one chunk)
cd "$(dirname "$0")/.."
# (the runtime reads it when the profiler's preloaded library initialises it -- before the program's own os.environ.setdefault runs)
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp LENTIL_STREAM=0 LENTIL_CHUNKS=1
TAG=${1:-x}; O=gpurun_out/pmc_accept_$TAG; mkdir -p $O
B="bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-second-regime --no-configs --no-pcie --no-parity-check --no-scan-alone"
i=1
for S in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU" \
         "SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  rm -rf /tmp/pa_$i
  timeout 600 rocprofv3 --pmc $S --output-format csv -d /tmp/pa_$i -- python3 $B > $O/set$i.log 2>&1
  f=$(find /tmp/pa_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" > $O/set$i.json <<'PY'
import csv, json, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "accept_kernel" not in k: continue
    # first-round launches only: the big ones
    acc[k[:40]][r["Counter_Name"]] += float(r["Counter_Value"]); n[k[:40]].add(r.get("Dispatch_Id"))
print(json.dumps({k: dict(v, launches=len(n[k])) for k, v in acc.items()}, indent=1))
PY
  i=$((i+1))
done
cat $O/set*.json
