#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s27; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
B="python3 bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-second-regime --no-configs --no-pcie --no-parity-check --no-scan-alone"
for rep in 1 2 3 4; do
    echo -n "rep $rep -> "
    $B 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["kernels_ms"], d["roofline"]["frac"], d["passes"]["chunks_redone_after_a_short_estimate"], d["passes"]["first_batch_model"]["lean_passes_lost"], d["box"]["fp64_mul_add_tflops"])'
done > $O/headline.txt 2>&1
bash tools/pass_sequence.sh > $O/pass_sequence.txt 2>&1
python3 tools/loop_overhead.py > $O/loop_overhead.txt 2>&1
timeout 600 python3 -m pytest tests/test_gpu_batch_model.py tests/test_gpu_headline.py -q -x -k "batch or second_round or headline_4k" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
