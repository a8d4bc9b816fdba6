#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s11; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
B="python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-second-regime --no-configs --no-pcie --no-parity-check --no-scan-alone"
for rep in 1 2; do
for pct in 100 90 82 75 68; do
  for pr in 0 1; do
    echo -n "rep $rep pct $pct predict $pr -> "
    LENTIL_SCAN_CUS_PCT=$pct LENTIL_PREDICT=$pr $B 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["kernels_ms"], d["passes"]["solve_accept_rounds_max"], d["passes"]["first_batch_model"])'
  done
done
done > $O/scan_cus.txt 2>&1
