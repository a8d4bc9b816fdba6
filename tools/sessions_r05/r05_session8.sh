#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s8; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
python3 tools/batch_model_check.py tools/data/item_need_double_gauss_50mm_5eed.npz tools/data/item_need_double_gauss_50mm_beef.npz tools/data/item_need_petzval_58mm_5eed.npz > $O/model_check.txt 2>&1
for v in "LENTIL_PREDICT=0" "LENTIL_PREDICT=1" "LENTIL_PREDICT=1 LENTIL_SCAN_OUTSIDE_IN=1" "LENTIL_PREDICT=1 LENTIL_SOLVE_B=1"; do
  echo "=== $v"; env $v python3 tools/per_pass.py 24 2>&1 | grep -v amdgpu.ids | tail -7
done > $O/variants_headline.txt 2>&1
for v in "LENTIL_PREDICT=0" "LENTIL_PREDICT=1"; do
  echo "=== $v"; env $v python3 tools/per_pass.py 16 petzval_58mm 8 2>&1 | grep -v amdgpu.ids | tail -9
done > $O/variants_config4.txt 2>&1
