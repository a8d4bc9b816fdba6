#!/bin/bash
# tight first batches for the items the model certifies (LENTIL_TIGHT_BATCHES=1: exactly `samples` traces) and no spare attempts
# (LENTIL_EXTRA_CONST=0) against the plain batch (samples + retries + 16): A/B in one process, lost bets counted; then parity with it on
cd "$(dirname "$0")/../.."
O=gpurun_out/r06s35; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
timeout 900 python3 tools/ab_inproc.py --reps 5 --steps 40 "" "LENTIL_EXTRA_CONST=0" "LENTIL_TIGHT_BATCHES=1" "LENTIL_TIGHT_BATCHES=1 LENTIL_EXTRA_CONST=0" > $O/ab_headline.txt 2>&1
tail -5 $O/ab_headline.txt; grep "rounds 2\|rounds 3" $O/ab_headline.txt | head -5
timeout 600 python3 tools/ab_inproc.py --reps 3 --steps 30 --lens petzval_58mm --aovs 8 "" "LENTIL_TIGHT_BATCHES=1" > $O/ab_config4.txt 2>&1
tail -3 $O/ab_config4.txt
timeout 600 python3 tools/ab_inproc.py --reps 3 --steps 40 --width 1920 --height 1080 --samples 256 "" "LENTIL_TIGHT_BATCHES=1" > $O/ab_config2.txt 2>&1
tail -3 $O/ab_config2.txt
LENTIL_TIGHT_BATCHES=1 timeout 900 python3 -m pytest tests/test_gpu_batch_model.py tests/test_gpu_async.py tests/test_gpu_headline.py -x -q -k "batch or pipelined or headline_4k or config2 or config3" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
