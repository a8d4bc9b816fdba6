#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s54; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
B="python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-second-regime --no-configs --no-pcie --no-parity-check --no-scan-alone"
run() { echo -n "$1 -> "; env $1 $B 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["kernels_ms"], d["passes"]["chunks_redone_after_a_short_estimate"], d["passes"]["first_batch_model"]["lean_passes_lost"], d["solve_fp64"]["scan_dominated"]["parked_solves_per_step"])'; }
for rep in 1 2; do
  run "X=0"
  run "LENTIL_SLOW_AT=12"
  run "LENTIL_SLOW_AT=30"
  run "LENTIL_SLOW_MAX_LANES=2"
  run "LENTIL_SLOW_MAX_LANES=8"
  run "LENTIL_SLOW_AT=12 LENTIL_SLOW_WAVES_PER_CU=2"
done > $O/ab.txt 2>&1
