#!/bin/bash
# (1) tools/micro/vgpr_alloc: does a SIMD hand a finished wave's registers to a new wave while older waves are resident?
# (2) scan_solve_kernel (the scan's waves go on as solve waves) against the oracle; (3) what it buys: config 4, the headline, config 5
cd "$(dirname "$0")/../.."
O=gpurun_out/r06s21; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
timeout 120 tools/micro/vgpr_alloc > $O/vgpr_alloc.txt 2>&1; echo "rc=$?" >> $O/vgpr_alloc.txt
timeout 1500 python3 -m pytest tests/test_gpu_fused_scan.py -x -q -s > $O/pytest_fused.log 2>&1; echo "rc=$?" >> $O/pytest_fused.log
tail -5 $O/pytest_fused.log
timeout 600 python3 tools/ab_inproc.py --reps 4 --steps 30 --lens petzval_58mm --aovs 8 "LENTIL_FUSED_SCAN=0" "LENTIL_FUSED_SCAN=1" > $O/ab_config4.txt 2>&1
tail -3 $O/ab_config4.txt
timeout 600 python3 tools/ab_inproc.py --reps 6 --steps 40 "LENTIL_FUSED_SCAN=0" "LENTIL_FUSED_SCAN=2" "LENTIL_FUSED_SCAN=2 LENTIL_SCAN_CUS_PCT=100" > $O/ab_headline.txt 2>&1
tail -4 $O/ab_headline.txt
timeout 600 python3 tools/ab_inproc.py --reps 3 --steps 12 --width 7680 --height 4320 --samples 2048 "LENTIL_FUSED_SCAN=0" "LENTIL_FUSED_SCAN=2" > $O/ab_config5.txt 2>&1
tail -3 $O/ab_config5.txt
