#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s38; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
PASSES=2 bash tools/pass_sequence.sh > $O/pass_sequence_two.txt 2>&1
B="python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-second-regime --no-configs --no-pcie --no-parity-check --no-scan-alone"
run() { echo -n "$1 -> "; env $1 $B 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["kernels_ms"], d["passes"]["chunks_redone_after_a_short_estimate"], d["passes"]["first_batch_model"]["lean_passes_lost"])'; }
for rep in 1 2 3; do
  run "X=0"
  run "LENTIL_SLOW_WAVES_PER_CU=2"
  run "LENTIL_SLOW_WAVES_PER_CU=3"
  run "LENTIL_SLOW_WAVES_PER_CU=4"
  run "LENTIL_SLOW_PRIO=3"
  run "LENTIL_SLOW_WAVES_PER_CU=2 LENTIL_SLOW_PRIO=3"
  run "LENTIL_SLOW_WAVES_PER_CU=4 LENTIL_SLOW_PRIO=3"
done > $O/ab.txt 2>&1
LENTIL_SLOW_WAVES_PER_CU=4 PASSES=2 bash tools/pass_sequence.sh > $O/pass_sequence_two_slow4.txt 2>&1
timeout 900 python3 -m pytest tests/test_gpu_batch_model.py tests/test_gpu_headline.py tests/test_gpu_parity.py -q -x -k "batch or headline_4k or second_round or stall or straggl or slow" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
