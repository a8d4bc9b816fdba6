#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s41; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
B="python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-second-regime --no-configs --no-pcie --no-parity-check --no-scan-alone"
run() { echo -n "$1 | $2 -> "; env $1 $B $2 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["kernels_ms"], d["passes"]["chunks_redone_after_a_short_estimate"], d["passes"]["first_batch_model"]["lean_passes_lost"])'; }
C4="--lens petzval_58mm --aovs 8"
for rep in 1 2 3; do
  run "X=0" "$C4"
  run "LENTIL_RESOLVE_AFTER_SCAN=1" "$C4"
  run "LENTIL_RESOLVE_AFTER_SCAN=1 LENTIL_EARLY_RESOLVE_BLOCKS=2" "$C4"
  run "LENTIL_EARLY_RESOLVE_BLOCKS=2" "$C4"
done > $O/ab.txt 2>&1
LENTIL_RESOLVE_AFTER_SCAN=1 bash tools/pass_sequence.sh $C4 > $O/pass_sequence_c4_ras.txt 2>&1
