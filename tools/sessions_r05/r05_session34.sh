#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s35; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
for rep in $(seq 1 14); do
  LENTIL_STREAM_DEBUG=1 timeout 600 python3 -m pytest tests/test_gpu_batch_model.py -q -s > $O/pytest_$rep.log 2>&1; echo "rep $rep rc=$?" >> $O/summary.txt
  grep -h "passed\|failed" $O/pytest_$rep.log | tail -1 >> $O/summary.txt
  grep -h "\[stream\] note\|\[stream\] redo\|\[stream\] queues\|\[stream\] stragglers\|\[stream\] scan" $O/pytest_$rep.log >> $O/summary.txt
done
