#!/bin/bash
# scan_solve_kernel with instance A running the same kernel (one copy of the solve loop on the chip): tests, config 4 A/B; and what
# two solve waves per SIMD deliver ALONE (the chunked pass of the highlight-heavy frame with LENTIL_SOLVE_BLOCKS=2 / 3 / 4)
cd "$(dirname "$0")/../.."
O=gpurun_out/r06s22; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
timeout 1500 python3 -m pytest tests/test_gpu_fused_scan.py -x -q -s > $O/pytest_fused.log 2>&1; echo "rc=$?" >> $O/pytest_fused.log
tail -5 $O/pytest_fused.log
timeout 600 python3 tools/ab_inproc.py --reps 4 --steps 30 --lens petzval_58mm --aovs 8 "LENTIL_FUSED_SCAN=0" "LENTIL_FUSED_SCAN=1" "LENTIL_FUSED_SCAN=1 LENTIL_FUSED_A=0" > $O/ab_config4.txt 2>&1
tail -4 $O/ab_config4.txt
timeout 600 python3 tools/ab_inproc.py --reps 2 --steps 3 --warmup 1 --f-hi 1.6e-3 --lens petzval_58mm "LENTIL_SOLVE_BLOCKS=2" "LENTIL_SOLVE_BLOCKS=3" "LENTIL_SOLVE_BLOCKS=4" > $O/ab_heavy_blocks_petzval.txt 2>&1
tail -4 $O/ab_heavy_blocks_petzval.txt
timeout 600 python3 tools/ab_inproc.py --reps 2 --steps 3 --warmup 1 --f-hi 1.6e-3 "LENTIL_SOLVE_BLOCKS=2" "LENTIL_SOLVE_BLOCKS=3" "LENTIL_SOLVE_BLOCKS=4" > $O/ab_heavy_blocks_dgauss.txt 2>&1
tail -4 $O/ab_heavy_blocks_dgauss.txt
timeout 600 python3 tools/ab_inproc.py --reps 4 --steps 40 "LENTIL_FUSED_SCAN=0" "LENTIL_FUSED_SCAN=2" > $O/ab_headline.txt 2>&1
tail -3 $O/ab_headline.txt
