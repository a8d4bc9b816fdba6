#!/bin/bash
# round 5, first GPU session: the review fixes' tests, a baseline of the headline step on this box, two switches, pass sequences
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s1; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "widest_records or petzval_8_aovs or extra_aovs" > $O/pytest_wide.log 2>&1; echo "pytest wide rc=$?" >> $O/pytest_wide.log
timeout 900 python3 -m pytest tests/test_plugin.py -q -x -k "moving_camera" > $O/pytest_plugin.log 2>&1; echo "pytest plugin rc=$?" >> $O/pytest_plugin.log
B="python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-second-regime --no-configs --no-pcie --no-parity-check --no-scan-alone"
$B 2>$O/bench_base.err | tail -1 > $O/bench_base.json
bash tools/ab_env.sh "LENTIL_SOLVE_B=0" "LENTIL_SOLVE_B=1" 2 > $O/ab_solve_b.txt 2>&1
bash tools/ab_env.sh "LENTIL_EXTEND=0" "LENTIL_EXTEND=1" 2 > $O/ab_extend.txt 2>&1
bash tools/ab_env.sh "LENTIL_ACCEPT_BLOCKS=2" "LENTIL_ACCEPT_BLOCKS=4" 2 > $O/ab_accept_blocks.txt 2>&1
bash tools/pass_sequence.sh > $O/pass_sequence_default.txt 2>&1
LENTIL_EXTEND=1 bash tools/pass_sequence.sh > $O/pass_sequence_extend.txt 2>&1
bash tools/clock_probe.sh > $O/clock_probe.txt 2>&1
