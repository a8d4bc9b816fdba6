#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/dev_${1:-x}; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_cases.py tests/test_gpu_headline.py -m gpu -x -q -k "po_redistribute_parity or direct_accumulation or decision_branches or headline_4k or blind_passes or randomized_configurations or full_size or config2 or config5_8k_bands or enable_dof or ragged" 2>&1 | tail -8 > $O/tests.log
cat $O/tests.log
bash tools/ab_bench.sh $1 pota_amd/_ab/liblentil_hip_r03.so pota_amd/liblentil_hip.so 2>&1 | tail -6
B="python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-second-regime --no-configs --no-pcie --no-parity-check --no-scan-alone"
P='import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["kernels_ms"], "frac", d["roofline"]["frac"], d["passes"]["streamed"], d["passes"]["solve_accept_rounds_max"])'
run() { echo -n "$* -> "; env "$@" 2>/dev/null | tail -1 | python3 -c "$P"; }
run X=1 $B --f-hi 0
run LENTIL_STREAM=0 LENTIL_CHUNKS=1 $B
run LENTIL_SCAN_DMA2=0 $B
