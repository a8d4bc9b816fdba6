#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s10; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
python3 tools/batch_model_check.py tools/data/item_need_double_gauss_50mm_5eed.npz tools/data/item_need_petzval_58mm_5eed.npz > $O/model_check.txt 2>&1
timeout 1200 python3 -m pytest tests/test_gpu_batch_model.py -q > $O/pytest_bm.log 2>&1; echo "rc=$?" >> $O/pytest_bm.log
bash tools/ab_env.sh "LENTIL_PREDICT=0" "LENTIL_PREDICT=1" 4 > $O/ab_predict.txt 2>&1
for v in "LENTIL_PREDICT=0" "LENTIL_PREDICT=1"; do
  echo "=== $v"; env $v python3 tools/per_pass.py 24 petzval_58mm 8 2>&1 | grep -v amdgpu.ids | tail -4
done > $O/variants_config4.txt 2>&1
