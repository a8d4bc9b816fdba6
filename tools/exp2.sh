#!/bin/bash
cd "$(dirname "$0")/.."
for b in 1 2 3 4; do echo "LENTIL_SOLVE_BLOCKS=$b"; LENTIL_SOLVE_BLOCKS=$b timeout 300 python3 tools/solve_workload.py double_gauss_50mm 1.6e-3 2 2>&1 | tail -1 | cut -c1-330; done
