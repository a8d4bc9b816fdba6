#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s16; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
