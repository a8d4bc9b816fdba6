#!/bin/bash
# a sweep of the knobs round 5 set, on round 6's pass: scanning CUs, straggler waves per CU, first-accept blocks per CU
cd "$(dirname "$0")/../.."
O=gpurun_out/r06s42; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
timeout 900 python3 tools/ab_inproc.py --reps 4 --steps 40 "" "LENTIL_SCAN_CUS_PCT=76" "LENTIL_SCAN_CUS_PCT=92" "LENTIL_SCAN_CUS_PCT=100" "LENTIL_SLOW_WAVES_PER_CU=2" "LENTIL_READY_BLOCKS=3" "LENTIL_READY_BLOCKS=5" "LENTIL_PUBLISH_WAVES=128" > $O/ab_sweep.txt 2>&1
tail -9 $O/ab_sweep.txt
