#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s44; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
# the item-counted end markers with the second round beside the first accept again: does it ever stall?
for rep in $(seq 1 16); do
  LENTIL_OVERLAP_ACCEPT=1 LENTIL_STREAM_DEBUG=1 timeout 600 python3 -m pytest tests/test_gpu_batch_model.py -q -s > $O/pytest_$rep.log 2>&1; echo "rep $rep rc=$?" >> $O/summary.txt
  grep -h "passed\|failed" $O/pytest_$rep.log | tail -1 >> $O/summary.txt
  grep -h "\[stream\] note" $O/pytest_$rep.log >> $O/summary.txt
done
for rep in 1 2 3 4; do
  LENTIL_OVERLAP_ACCEPT=1 LENTIL_STREAM_DEBUG=1 timeout 600 python3 -m pytest tests/test_gpu_headline.py -q -s -k headline_4k > $O/headline_$rep.log 2>&1; echo "headline rep $rep rc=$?" >> $O/summary.txt
  grep -h "passed\|failed" $O/headline_$rep.log | tail -1 >> $O/summary.txt
  grep -h "\[stream\] note" $O/headline_$rep.log >> $O/summary.txt
done
# both orders: the tests that know about stalls, blind passes and fallbacks
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_batch_model.py -q -x -k "stall or blind or batch or second_round or hardware_queues" > $O/pytest_default.log 2>&1; echo "rc=$?" >> $O/pytest_default.log
LENTIL_OVERLAP_ACCEPT=1 timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "stall or blind or hardware_queues" > $O/pytest_overlap.log 2>&1; echo "rc=$?" >> $O/pytest_overlap.log
# what the order costs: config 5's bands (a second round in every pass)
P='import sys,json; d=json.loads(sys.stdin.read()); print("%.3f ms" % d["ms_per_step"], d.get("kernels_ms"), "streamed %s redone %s rounds %s" % (d["passes"]["streamed"], d["passes"]["chunks_redone_after_a_short_estimate"], d["passes"]["solve_accept_rounds_max"]))'
C="--no-cpu-baseline --no-configs --no-pcie --no-second-regime --no-parity-check --no-scan-alone --steps 12 --warmup 3 --width 7680 --height 4320 --samples 2048"
for rep in 1 2; do
for e in "8,0" "8,3" "4,1"; do
  for ov in 0 1; do echo -n "overlap $ov N,r=$e : "; LENTIL_OVERLAP_ACCEPT=$ov timeout 300 python3 bench.py $C --emulate $e 2>/dev/null | tail -1 | python3 -c "$P"; done
done
done > $O/bands.txt 2>&1
