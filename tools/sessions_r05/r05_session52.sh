#!/bin/bash
cd "$(dirname "$0")/../.."
export GPU_MAX_HW_QUEUES=8
python3 -m pytest tests -m gpu -q -x > gpurun_out/r05_full_suite_final2.log 2>&1; echo "rc=$?" >> gpurun_out/r05_full_suite_final2.log
mkdir -p gpurun_out/r05/profiles
python3 bench.py --steps 20 --warmup 3 > gpurun_out/r05/profiles/r05_bench_line.json 2> gpurun_out/r05/bench.err
PASSES=1 bash tools/pass_sequence.sh > gpurun_out/r05/profiles/r05_pass_sequence.txt 2>&1
