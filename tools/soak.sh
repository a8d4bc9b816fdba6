#!/bin/bash
# The seeded soaks, randomised sweeps and child-process runs of the GPU suite (marker gpu_soak: tests/test_soak_gpu.py; plus the
# soaks that live beside their fixed cases) under other seeds / more cases than the defaults: tools/soak.sh <seed> <cases>
cd "$(dirname "$0")/.."
export LENTIL_SOAK_SEED=${1:-0xA11CE} LENTIL_SOAK_CASES=${2:-24}
timeout 2400 python -m pytest tests -m "gpu_soak or gpu" -x -q -k "soak or randomized or few_hardware_queues" 2>&1 | tail -8
