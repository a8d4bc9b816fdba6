#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s3; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
timeout 1200 python3 -m pytest tests/test_gpu_batch_model.py -q > $O/pytest_bm.log 2>&1; echo "rc=$?" >> $O/pytest_bm.log
bash tools/ab_env.sh "LENTIL_PREDICT=0" "LENTIL_PREDICT=1" 3 > $O/ab_predict.txt 2>&1
B="python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-second-regime --no-configs --no-pcie --no-parity-check --no-scan-alone"
$B 2>$O/bench.err | tail -1 > $O/bench.json
bash tools/pass_sequence.sh > $O/pass_sequence.txt 2>&1
bash tools/ab_env.sh "LENTIL_ACCEPT_BLOCKS=2" "LENTIL_ACCEPT_BLOCKS=4" 2 > $O/ab_accept_blocks.txt 2>&1
bash tools/ab_env.sh "LENTIL_EARLY_RESOLVE_BLOCKS=4" "LENTIL_EARLY_RESOLVE_BLOCKS=8" 2 > $O/ab_resolve_blocks.txt 2>&1
