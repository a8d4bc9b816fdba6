#!/bin/bash
# accept_kernel<5> (LENTIL_EARLY_ACCEPT=2): the first accept in item order beside the solve kernel's ramp-down -- results written
# through and counted without a wait, the window re-read while it still holds the pool's wipe value.  Parity, then A/B.
cd "$(dirname "$0")/../.."
O=gpurun_out/r06s36; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
LENTIL_EARLY_ACCEPT=2 LENTIL_STREAM_DEBUG=1 timeout 900 python3 -m pytest tests/test_gpu_batch_model.py tests/test_gpu_async.py tests/test_gpu_headline.py -x -q -k "batch or pipelined or headline_4k or config2" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log; grep -h "\[stream\] note" $O/pytest.log | head -3 | cut -c1-400
timeout 900 python3 tools/ab_inproc.py --reps 5 --steps 40 "" "LENTIL_EARLY_ACCEPT=2" "LENTIL_EARLY_ACCEPT=2 LENTIL_EARLY_ACCEPT_BLOCKS=2" "LENTIL_EARLY_ACCEPT=2 LENTIL_EARLY_ACCEPT_BLOCKS=4" "LENTIL_EARLY_ACCEPT=1" > $O/ab_headline.txt 2>&1
tail -6 $O/ab_headline.txt
timeout 600 python3 tools/ab_inproc.py --reps 3 --steps 40 --width 1920 --height 1080 --samples 256 "" "LENTIL_EARLY_ACCEPT=2" > $O/ab_config2.txt 2>&1
tail -3 $O/ab_config2.txt
PASSES=1 LENTIL_EARLY_ACCEPT=2 bash tools/pass_sequence.sh > $O/pass_sequence_early2.txt 2>&1
head -16 $O/pass_sequence_early2.txt
