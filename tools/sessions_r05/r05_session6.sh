#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s6; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
export LENTIL_PREDICT=1 LENTIL_SCAN_OUTSIDE_IN=1 LENTIL_PARK_DRY_ONLY=0
PASSES=4 bash tools/pass_sequence.sh > $O/pass_sequence4.txt 2>&1
python3 tools/timeline.py --passes 8 > $O/timeline.txt 2>&1
