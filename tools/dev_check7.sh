#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/dev_${1:-x}; mkdir -p $O
B="python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-second-regime --no-configs --no-pcie --no-parity-check --no-scan-alone"
P='import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["kernels_ms"], "frac", d["roofline"]["frac"], d["passes"])'
run() { echo -n "$* -> "; timeout 300 env "$@" 2>/dev/null | tail -1 | python3 -c "$P"; }
run LENTIL_EXTEND=0 $B
run LENTIL_EXTEND=1 LENTIL_LEAN_TAIL=0 $B
run LENTIL_EXTEND=1 LENTIL_LEAN_TAIL=1 $B
run LENTIL_EXTEND=1 LENTIL_LEAN_TAIL=1 $B
LENTIL_STREAM_DEBUG=1 timeout 300 python3 tools/timeline.py --passes 7 --out $O/timeline.txt 2>&1 | grep -E "stream\]|pass [0-9]" | head; grep -A16 "kernel spans" $O/timeline.txt | cut -c1-90
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline.py -m gpu -x -q -k "blind or stragglers or randomized_configurations or headline_4k or chromatic_streamed or config4_like or po_redistribute_parity" 2>&1 | tail -8
