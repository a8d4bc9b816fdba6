#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s49; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
timeout 900 python3 -m pytest tests/test_crypto.py -m gpu -q -x > $O/pytest_crypto.log 2>&1; echo "rc=$?" >> $O/pytest_crypto.log
for rep in 1 2 3 4 5 6; do
  timeout 600 python3 -m pytest tests/test_gpu_batch_model.py -q -k "second_round" > $O/pytest_$rep.log 2>&1; echo "rep $rep rc=$?" >> $O/summary.txt
  grep -h "AssertionError\|passed\|failed" $O/pytest_$rep.log | cut -c1-600 >> $O/summary.txt
done
