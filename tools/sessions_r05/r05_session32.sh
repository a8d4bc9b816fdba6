#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s32; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
for rep in 1 2 3 4 5 6 7 8 9 10; do
  timeout 600 python3 -m pytest tests/test_gpu_batch_model.py tests/test_gpu_headline.py -q -k "batch or second_round or headline_4k" > $O/pytest_$rep.log 2>&1; echo "rep $rep rc=$?" >> $O/summary.txt
  grep -h "AssertionError: \|passed\|failed" $O/pytest_$rep.log >> $O/summary.txt
done
timeout 900 python3 tools/first_streamed_soak.py 40 0 > $O/soak_busy.txt 2>&1
timeout 900 python3 tools/first_streamed_soak.py 20 1.5 > $O/soak_idle.txt 2>&1
