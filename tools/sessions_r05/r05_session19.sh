#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s19; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
B="python3 bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-second-regime --no-configs --no-pcie --no-parity-check --no-scan-alone"
for rep in 1 2 3; do
for v in "LENTIL_ACCEPT_BLOCKS=2" "LENTIL_ACCEPT_BLOCKS=3" "LENTIL_ACCEPT_BLOCKS=4" "LENTIL_EARLY_RESOLVE_BLOCKS=8" "LENTIL_SCAN_OUTSIDE_IN=1"; do
    echo -n "rep $rep $v -> "
    env $v $B 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["kernels_ms"], d["passes"]["chunks_redone_after_a_short_estimate"], d["passes"]["first_batch_model"]["lean_passes_lost"])'
done
done > $O/tail_knobs.txt 2>&1
bash tools/pass_sequence.sh > $O/pass_sequence.txt 2>&1
