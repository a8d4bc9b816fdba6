#!/bin/bash
# PMC passes over the solve kernels on a highlight-heavy frame (chunked form).  usage: tools/pmc_solve.sh <tag> [lenses...]
# Each rocprofv3 run is its own pass (SQ has 8 slots); the program comes straight after `--`.
cd "$(dirname "$0")/.."
# (the runtime reads it when the profiler's preloaded library initialises it -- before the program's own os.environ.setdefault runs)
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
TAG=${1:-r04}; shift
LENSES=${@:-"double_gauss_50mm petzval_58mm"}
O=gpurun_out/pmc_$TAG; mkdir -p $O
rocprofv3 -L > $O/counters_avail.txt 2>&1
SET1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
SET2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS"
SET3="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_VALU_ADD_F64"
for L in $LENSES; do
  i=1
  for S in "$SET1" "$SET2" "$SET3"; do
    rm -rf /tmp/pmc_${L}_$i
    timeout 900 rocprofv3 --pmc $S --output-format csv -d /tmp/pmc_${L}_$i -- python3 tools/solve_workload.py $L 1.6e-3 2 > $O/${L}_set$i.log 2>&1
    f=$(find /tmp/pmc_${L}_$i -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python3 tools/pmc_solve_summary.py "$f" > $O/${L}_set$i.json
    i=$((i+1))
  done
  timeout 600 python3 tools/solve_workload.py $L 1.6e-3 3 > $O/${L}_unprofiled.log 2>&1
done
