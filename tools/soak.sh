#!/bin/bash
# The seeded soak tests of the GPU suite under other seeds / more cases than the defaults: tools/soak.sh <seed> <cases>
cd "$(dirname "$0")/.."
export LENTIL_SOAK_SEED=${1:-0xA11CE} LENTIL_SOAK_CASES=${2:-24}
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_multi_gpu.py tests/test_native_exchange.py -m gpu -x -q -k "soak or randomized" 2>&1 | tail -8
