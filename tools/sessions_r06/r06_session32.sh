#!/bin/bash
# small tasks at the end of the queue (LENTIL_SMALL_TAIL_PCT / _UNITS): A/B in one process, then parity with it on
cd "$(dirname "$0")/../.."
O=gpurun_out/r06s32; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
timeout 900 python3 tools/ab_inproc.py --reps 5 --steps 40 "" "LENTIL_SMALL_TAIL_PCT=10" "LENTIL_SMALL_TAIL_PCT=25" "LENTIL_SMALL_TAIL_PCT=25 LENTIL_SMALL_TAIL_UNITS=32" "LENTIL_SMALL_TAIL_PCT=50 LENTIL_SMALL_TAIL_UNITS=32" "LENTIL_SMALL_TAIL_PCT=100 LENTIL_SMALL_TAIL_UNITS=32" > $O/ab_headline.txt 2>&1
tail -7 $O/ab_headline.txt
timeout 600 python3 tools/ab_inproc.py --reps 3 --steps 30 --lens petzval_58mm --aovs 8 "" "LENTIL_SMALL_TAIL_PCT=10" "LENTIL_SMALL_TAIL_PCT=25" > $O/ab_config4.txt 2>&1
tail -4 $O/ab_config4.txt
LENTIL_SMALL_TAIL_PCT=25 timeout 900 python3 -m pytest tests/test_gpu_batch_model.py tests/test_gpu_async.py tests/test_gpu_headline.py -x -q -k "batch or pipelined or headline_4k or config2" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
