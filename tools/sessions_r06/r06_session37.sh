#!/bin/bash
# the GPU suite as the driver runs it, on the round's last commit; smoke() before it
cd "$(dirname "$0")/../.."
O=gpurun_out/r06s37; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
S=$(date +%s)
python3 -m pytest tests -m gpu -q -x > $O/r06_gpu_suite_final_commit.log 2>&1; echo "rc=$? wall=$(( $(date +%s) - S )) s" >> $O/r06_gpu_suite_final_commit.log
tail -3 $O/r06_gpu_suite_final_commit.log
