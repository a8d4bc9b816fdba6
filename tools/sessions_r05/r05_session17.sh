#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s17; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
timeout 900 python3 -m pytest tests/test_gpu_lens_jit.py "tests/test_gpu_parity.py::test_streamed_pass_that_stalls_after_its_first_accept_is_run_again" -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
B="python3 bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-second-regime --no-configs --no-pcie --no-parity-check --no-scan-alone"
for rep in 1 2 3 4; do
for pct in 100 92 84 76; do
    echo -n "rep $rep pct $pct -> "
    LENTIL_SCAN_CUS_PCT=$pct $B 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["kernels_ms"], d["passes"]["chunks_redone_after_a_short_estimate"], d["passes"]["first_batch_model"]["lean_passes_lost"], d["box"]["fp64_mul_add_tflops"])'
done
done > $O/scan_cus.txt 2>&1
