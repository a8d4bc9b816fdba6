#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s12; mkdir -p $O
export GPU_MAX_HW_QUEUES=8 LENTIL_STREAM_DEBUG=1
B="python3 bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-second-regime --no-configs --no-pcie --no-parity-check --no-scan-alone"
for rep in 1 2 3 4 5 6 7 8; do
  for pr in 0 1; do
    echo -n "rep $rep predict $pr -> "
    LENTIL_PREDICT=$pr $B 2>$O/err_${pr}_$rep.txt | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["kernels_ms"], d["passes"])'
    grep -h "\[stream\]" $O/err_${pr}_$rep.txt | head -8
  done
done > $O/stalls.txt 2>&1
