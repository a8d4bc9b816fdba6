#!/bin/bash
# the second round beside the first accept as the default: the tests that keep a second round in flight, in both orders
cd "$(dirname "$0")/../.."
O=gpurun_out/r06s27; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
timeout 1500 python3 -m pytest tests/test_gpu_batch_model.py tests/test_gpu_async.py tests/test_gpu_parity.py -x -q -k "batch or second_round or stall or blind or hardware_queues or pipelined or deferred or two_passes or streamed" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
