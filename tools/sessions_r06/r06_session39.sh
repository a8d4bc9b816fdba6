#!/bin/bash
# the end of a deferred pass handed to the context's stream by a word in device memory (pass_done_kernel / pass_wait_kernel,
# LENTIL_FLAG_HOP) instead of an event: A/B in one process, the tests of the asynchronous end, two passes' kernel sequence
cd "$(dirname "$0")/../.."
O=gpurun_out/r06s39; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
timeout 900 python3 tools/ab_inproc.py --reps 6 --steps 40 "LENTIL_FLAG_HOP=0" "LENTIL_FLAG_HOP=1" > $O/ab_headline.txt 2>&1
tail -3 $O/ab_headline.txt
timeout 600 python3 tools/ab_inproc.py --reps 4 --steps 40 --width 1920 --height 1080 --samples 256 "LENTIL_FLAG_HOP=0" "LENTIL_FLAG_HOP=1" > $O/ab_config2.txt 2>&1
tail -3 $O/ab_config2.txt
timeout 900 python3 -m pytest tests/test_gpu_async.py tests/test_gpu_batch_model.py tests/test_gpu_headline.py -x -q -k "async or pipelined or deferred or two_passes or batch or headline_4k" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
PASSES=2 bash tools/pass_sequence.sh > $O/pass_sequence_flag_hop.txt 2>&1
grep -v probe $O/pass_sequence_flag_hop.txt | head -34
