#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/dev_${1:-x}; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline.py -m gpu -x -q -k "blind or stragglers or randomized_configurations or headline_4k or chromatic_streamed or config4_like" 2>&1 | tail -5 > $O/tests.log
cat $O/tests.log
B="python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-second-regime --no-configs --no-pcie --no-parity-check --no-scan-alone"
P='import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["kernels_ms"], "frac", d["roofline"]["frac"], d["passes"]["streamed"], d["passes"]["solve_accept_rounds_max"])'
run() { echo -n "$* -> "; env "$@" 2>/dev/null | tail -1 | python3 -c "$P"; }
for r in 1 2 3; do run X=1 $B; done
run X=1 $B --lens petzval_58mm --aovs 8
python3 tools/timeline.py --passes 7 --out $O/timeline.txt > /dev/null 2>&1; grep -A14 "kernel spans" $O/timeline.txt | cut -c1-80
