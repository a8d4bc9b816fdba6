#!/bin/bash
cd "$(dirname "$0")/.."
B="python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-second-regime --no-configs --no-pcie --no-parity-check --no-scan-alone"
P='import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["kernels_ms"], "frac", d["roofline"]["frac"], d["passes"]["streamed"], d["passes"]["solve_accept_rounds_max"])'
run() { echo -n "$* -> "; env "$@" 2>/dev/null | tail -1 | python3 -c "$P"; }
run X=1 $B
run X=1 $B --f-hi 0
run X=1 $B --f-hi 1e-7
run LENTIL_STREAM_BLOCKS=1 $B
run LENTIL_DMA_BLOCKS=2 $B
run LENTIL_DMA_BLOCKS=2 LENTIL_STREAM_BLOCKS=1 $B
run LENTIL_DMA_RING=3 $B
run LENTIL_DMA_BLOCKS=2 $B --f-hi 0
run LENTIL_STREAM=0 LENTIL_CHUNKS=1 $B
run LENTIL_STREAM=0 LENTIL_CHUNKS=1 LENTIL_DMA_BLOCKS=1 $B
run X=1 $B
