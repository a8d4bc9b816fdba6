#!/bin/bash
# LENTIL_OVERLAP_ACCEPT=1 (the second round's resident kernels beside the first accept, rounds 3-4's order) soaked: passes that have a
# second round in flight -- config 5's bands (2 048 draws per item: no first-batch model), each alone on this GPU -- by the hundred,
# with the dispatch probe's build (a snapshot whenever the accept's last item is done with blocks of its grid not begun); then the
# tests that know about second rounds, looped.  What counts: passes redone / stuck (lentil_hip_process_stats through bench's line).
cd "$(dirname "$0")/../.."
O=gpurun_out/r06s26; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
P='import sys,json; d=json.loads(sys.stdin.read()); p=d["passes"]; print("%.3f ms" % d["ms_per_step"], "timed %s streamed %s redone %s rounds_max %s notes %s | process: %s" % (p["timed"], p["streamed"], p["chunks_redone_after_a_short_estimate"], p["solve_accept_rounds_max"], p.get("redo_notes"), p.get("process")))'
C="--no-cpu-baseline --no-configs --no-pcie --no-second-regime --no-parity-check --no-scan-alone --warmup 3 --width 7680 --height 4320 --samples 2048"
for ov in 1 0; do
  for e in "8,0" "8,3" "8,7" "4,1"; do
    echo -n "overlap $ov N,r=$e 300 steps: "
    LENTIL_HIP_LIB=$PWD/pota_amd/_ab/liblentil_hip_probe.so LENTIL_DISPATCH_PROBE=1 LENTIL_STREAM_DEBUG=1 LENTIL_OVERLAP_ACCEPT=$ov timeout 900 python3 bench.py $C --steps 300 --emulate $e 2>$O/err_${ov}_${e/,/_}.log | tail -1 | python3 -c "$P"
    grep -h "\[probe\]\|\[stream\] note" $O/err_${ov}_${e/,/_}.log | cut -c1-600 | head -5
  done
done > $O/bands_soak.txt 2>&1
cat $O/bands_soak.txt
# the headline's shape with the model off (a second round in every pass) and the whole 8K frame
for ov in 1 0; do
  echo -n "overlap $ov headline LENTIL_PREDICT=0 400 steps: "
  LENTIL_HIP_LIB=$PWD/pota_amd/_ab/liblentil_hip_probe.so LENTIL_DISPATCH_PROBE=1 LENTIL_STREAM_DEBUG=1 LENTIL_PREDICT=0 LENTIL_OVERLAP_ACCEPT=$ov timeout 900 python3 bench.py --no-cpu-baseline --no-configs --no-pcie --no-second-regime --no-parity-check --no-scan-alone --warmup 3 --steps 400 2>$O/err_headline_$ov.log | tail -1 | python3 -c "$P"
  grep -h "\[probe\]\|\[stream\] note" $O/err_headline_$ov.log | cut -c1-600 | head -5
done > $O/headline_soak.txt 2>&1
cat $O/headline_soak.txt
for rep in 1 2 3 4 5 6; do
  LENTIL_OVERLAP_ACCEPT=1 LENTIL_STREAM_DEBUG=1 timeout 600 python3 -m pytest tests/test_gpu_batch_model.py tests/test_gpu_async.py -q -x 2>&1 | grep -h "passed\|failed\|\[stream\] note" | cut -c1-400
done > $O/tests_looped.txt 2>&1
cat $O/tests_looped.txt
