#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s48; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
bash tools/soak.sh 0xBADC0DE 24 > $O/soak_1.txt 2>&1
bash tools/soak.sh 0x5EA5 24 > $O/soak_2.txt 2>&1
for rep in $(seq 1 12); do
  timeout 600 python3 -m pytest tests/test_gpu_batch_model.py tests/test_gpu_headline.py -q -k "batch or second_round or headline_4k" > $O/pytest_$rep.log 2>&1; echo "rep $rep rc=$?" >> $O/summary.txt
  grep -h "AssertionError: \|passed\|failed" $O/pytest_$rep.log >> $O/summary.txt
done
