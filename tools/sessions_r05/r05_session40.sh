#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s40; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
LENTIL_SOLVE_B=1 python3 tools/timeline.py --lens petzval_58mm --aovs 8 --passes 5 --out $O/timeline_c4_b.txt > $O/tl_b.log 2>&1
LENTIL_SOLVE_B=1 LENTIL_SOLVE_B_THREADS=256 python3 tools/timeline.py --lens petzval_58mm --aovs 8 --passes 5 --out $O/timeline_c4_b256.txt > $O/tl_b256.log 2>&1
python3 tools/timeline.py --lens petzval_58mm --aovs 8 --passes 5 --out $O/timeline_c4.txt > $O/tl.log 2>&1
