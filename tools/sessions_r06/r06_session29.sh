#!/bin/bash
# the scan from the frame's edges inwards (edge items' long solves first), now that the pass ends on the stragglers; and the spread of
# the headline line on one box (five short bench runs)
cd "$(dirname "$0")/../.."
O=gpurun_out/r06s29; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
timeout 600 python3 tools/ab_inproc.py --reps 6 --steps 40 "" "LENTIL_SCAN_OUTSIDE_IN=1" > $O/ab_outside_in.txt 2>&1
tail -3 $O/ab_outside_in.txt
for i in 1 2 3 4 5; do
  timeout 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-configs --no-second-regime --no-pcie --no-scan-alone --no-parity-check 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); b=d["roofline"]["box"]; print("%.4f ms  kernels %s  box %.1f TFLOP/s %.0f MHz before, %.0f MHz after" % (d["ms_per_step"], d["roofline"]["kernels_ms"], b["fp64_mul_add_tflops"], b["shader_clock_mhz_under_fp64"], b["after_the_timed_steps"]["shader_clock_mhz_under_fp64"]))'
done > $O/bench_spread.txt 2>&1
cat $O/bench_spread.txt
