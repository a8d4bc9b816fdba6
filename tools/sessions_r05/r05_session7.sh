#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s7; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
python3 tools/batch_model_check.py tools/data/item_need_double_gauss_50mm_5eed.npz tools/data/item_need_double_gauss_50mm_beef.npz tools/data/item_need_petzval_58mm_5eed.npz > $O/model_check.txt 2>&1
LENTIL_PREDICT=0 python3 tools/timeline.py --passes 8 > $O/timeline_nopredict.txt 2>&1
LENTIL_PREDICT=1 python3 tools/timeline.py --passes 8 > $O/timeline_predict.txt 2>&1
