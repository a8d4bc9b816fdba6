#!/bin/bash
# development round trip on the GPU box: a fast parity subset, the solve kernels' rate, the headline step
cd "$(dirname "$0")/.."
O=gpurun_out/dev_${1:-x}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "po_redistribute_parity or chromatic_aberration or stragglers or bokeh_image or blind_passes or randomized_configurations or sub_batches or petzval or polygonal or trace_bw" 2>&1 | tail -5 > $O/tests.log
cat $O/tests.log
for L in double_gauss_50mm petzval_58mm; do timeout 300 python3 tools/solve_workload.py $L 1.6e-3 3 2>&1 | tail -2 | cut -c1-400; done | tee $O/solve_rate.log
timeout 600 python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-second-regime --no-configs --no-pcie --no-parity-check 2>$O/bench.err | tail -1 > $O/bench.json
python3 - <<PY
import json
d=json.load(open("$O/bench.json")); print("headline ms", d["ms_per_step"], d.get("kernels_ms"), "roofline", d["roofline"]["frac"], d["roofline"].get("alone",{}).get("frac"), d["passes"])
PY
