#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s18; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
B="python3 bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-second-regime --no-configs --no-pcie --no-parity-check --no-scan-alone --lens petzval_58mm --aovs 8"
for rep in 1 2 3; do
for pct in 100 88 76 64; do
    echo -n "config4 rep $rep pct $pct -> "
    LENTIL_SCAN_CUS_PCT_MULTI=$pct $B 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["kernels_ms"], d["passes"]["chunks_redone_after_a_short_estimate"], d["passes"]["first_batch_model"]["lean_passes_lost"], d["box"]["fp64_mul_add_tflops"])'
done
done > $O/scan_cus_multi.txt 2>&1
# config 2 and config 5 with the new dma2 default against 100
for cfg in "--width 1920 --height 1080 --samples 256" "--width 7680 --height 4320 --samples 2048"; do
for pct in 100 84; do
  echo -n "$cfg pct $pct -> "
  LENTIL_SCAN_CUS_PCT=$pct python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-second-regime --no-configs --no-pcie --no-parity-check --no-scan-alone $cfg 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["kernels_ms"], d["passes"]["chunks_redone_after_a_short_estimate"], d["passes"]["solve_accept_rounds_max"])'
done
done > $O/other_configs.txt 2>&1
