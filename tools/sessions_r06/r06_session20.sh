#!/bin/bash
# The round's profile set on one box: tools/profile_round.sh r06 + pass sequences (headline, config 4) + cryptomatte sequence,
# then the GPU suite as the driver runs it (its wall time is what GPUTEST records).
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r06/profiles
bash tools/profile_round.sh r06 > gpurun_out/r06_profile_round.log 2>&1
PASSES=1 bash tools/pass_sequence.sh > gpurun_out/r06/profiles/r06_pass_sequence.txt 2>&1
bash tools/pass_sequence.sh --lens petzval_58mm --aovs 8 > gpurun_out/r06/profiles/r06_pass_sequence_config4.txt 2>&1
bash tools/crypto_sequence.sh > gpurun_out/r06/profiles/r06_crypto_sequence.txt 2>&1
export GPU_MAX_HW_QUEUES=8
S=$(date +%s)
python3 -m pytest tests -m gpu -q -x > gpurun_out/r06/profiles/r06_gpu_suite.log 2>&1; echo "rc=$? wall=$(( $(date +%s) - S )) s" >> gpurun_out/r06/profiles/r06_gpu_suite.log
tail -3 gpurun_out/r06/profiles/r06_gpu_suite.log
tail -c 1500 gpurun_out/r06/profiles/r06_bench_line.json
