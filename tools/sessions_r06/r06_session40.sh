#!/bin/bash
# the seeded soaks / randomised sweeps / child-process runs (marker gpu_soak) on the round's last commit, default seed and one more
cd "$(dirname "$0")/../.."
O=gpurun_out/r06s40; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
S=$(date +%s)
bash tools/soak.sh > $O/soak_default.txt 2>&1; echo "wall $(( $(date +%s) - S )) s" >> $O/soak_default.txt
tail -4 $O/soak_default.txt
S=$(date +%s)
bash tools/soak.sh 0xC0FFEE 16 > $O/soak_c0ffee.txt 2>&1; echo "wall $(( $(date +%s) - S )) s" >> $O/soak_c0ffee.txt
tail -4 $O/soak_c0ffee.txt
