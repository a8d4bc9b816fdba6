#!/bin/bash
# hard items first (LENTIL_HARD_FIRST: the tasks of the items the first-batch model enlarges on a queue of their own that every
# resident solve wave looks at at each task boundary): parity, then A/B in one process
cd "$(dirname "$0")/../.."
O=gpurun_out/r06s31; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
timeout 900 python3 -m pytest tests/test_gpu_batch_model.py tests/test_gpu_async.py -x -q > $O/pytest_small.log 2>&1; echo "rc=$?" >> $O/pytest_small.log
tail -4 $O/pytest_small.log
timeout 600 python3 tools/ab_inproc.py --reps 6 --steps 40 "LENTIL_HARD_FIRST=0" "LENTIL_HARD_FIRST=1" > $O/ab_headline.txt 2>&1
tail -3 $O/ab_headline.txt
timeout 600 python3 tools/ab_inproc.py --reps 4 --steps 30 --lens petzval_58mm --aovs 8 "LENTIL_HARD_FIRST=0" "LENTIL_HARD_FIRST=1" > $O/ab_config4.txt 2>&1
tail -3 $O/ab_config4.txt
timeout 600 python3 tools/ab_inproc.py --reps 4 --steps 40 --width 1920 --height 1080 --samples 256 "LENTIL_HARD_FIRST=0" "LENTIL_HARD_FIRST=1" > $O/ab_config2.txt 2>&1
tail -3 $O/ab_config2.txt
timeout 900 python3 -m pytest tests/test_gpu_headline.py -x -q -k "headline_4k or config2 or config4_like" > $O/pytest_headline.log 2>&1; echo "rc=$?" >> $O/pytest_headline.log
tail -4 $O/pytest_headline.log
timeout 300 python3 tools/timeline.py --passes 7 --out $O/timeline_headline_hard_first.txt > /dev/null 2>&1
