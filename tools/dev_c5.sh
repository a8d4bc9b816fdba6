#!/bin/bash
cd "$(dirname "$0")/.."
C5="bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-second-regime --no-configs --no-pcie --no-parity-check --no-scan-alone --width 7680 --height 4320 --samples 2048"
P='import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["kernels_ms"], d["passes"])'
for r in 1 2 3; do for e in 1 0; do echo -n "LENTIL_OVERLAP_HEAVY=$e -> "; LENTIL_OVERLAP_HEAVY=$e python3 $C5 2>/dev/null | tail -1 | python3 -c "$P"; done; done
H="bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-second-regime --no-configs --no-pcie --no-parity-check --no-scan-alone --f-hi 1.6e-3"
for e in 1 0; do echo -n "heavy regime LENTIL_OVERLAP_HEAVY=$e -> "; LENTIL_OVERLAP_HEAVY=$e python3 $H 2>/dev/null | tail -1 | python3 -c "$P"; done
timeout 1500 python -m pytest tests/test_gpu_headline.py -m gpu -x -q -k "config5" 2>&1 | tail -3
