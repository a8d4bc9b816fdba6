#!/bin/bash
# per-pixel splat flags (FrameDev::touched_px): the tests that compare resolved frames with the oracle, then old build against new
# on one box (tools/ab_bench.sh): the headline and config 4
cd "$(dirname "$0")/../.."
O=gpurun_out/r06s24; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
timeout 1500 python3 -m pytest tests/test_gpu_headline.py tests/test_gpu_async.py -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
bash tools/ab_bench.sh r06s24_headline pota_amd/_ab/liblentil_hip_base.so pota_amd/liblentil_hip.so > $O/ab_headline.txt 2>&1
cat $O/ab_headline.txt
bash tools/ab_bench.sh r06s24_config4 pota_amd/_ab/liblentil_hip_base.so pota_amd/liblentil_hip.so --lens petzval_58mm --aovs 8 > $O/ab_config4.txt 2>&1
cat $O/ab_config4.txt
PASSES=1 bash tools/pass_sequence.sh --lens petzval_58mm --aovs 8 > $O/pass_sequence_config4.txt 2>&1
PASSES=2 bash tools/pass_sequence.sh > $O/pass_sequence_headline.txt 2>&1
