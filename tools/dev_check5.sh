#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/dev_${1:-x}; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_cases.py tests/test_gpu_headline.py -m gpu -x -q -k "closest or debug or blind or stragglers or randomized_configurations or headline_4k or sub_batches or thinlens_redistribute or po_redistribute_parity or decision_branches or tiled_output_on_one" 2>&1 | tail -8 > $O/tests.log
cat $O/tests.log
B="python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-second-regime --no-configs --no-pcie --no-parity-check --no-scan-alone"
P='import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["kernels_ms"], "frac", d["roofline"]["frac"], d["passes"]["streamed"], d["passes"]["solve_accept_rounds_max"])'
run() { echo -n "$* -> "; env "$@" 2>/dev/null | tail -1 | python3 -c "$P"; }
for r in 1 2; do
run LENTIL_ACCEPT_WIDE=0 $B
run LENTIL_ACCEPT_WIDE=1 $B
run LENTIL_HIP_LIB=$PWD/pota_amd/_ab/liblentil_hip_eu4.so $B
done
run LENTIL_ACCEPT_WIDE=1 LENTIL_ACCEPT_BLOCKS=3 $B
run LENTIL_ACCEPT_WIDE=1 LENTIL_ACCEPT_BLOCKS=4 $B
python3 tools/timeline.py --passes 7 --out $O/timeline.txt > /dev/null 2>&1; grep -A14 "kernel spans" $O/timeline.txt | cut -c1-80; tail -1 $O/timeline.txt | cut -c1-260
