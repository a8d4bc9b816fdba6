#!/bin/bash
# The round's profile set on its final code, one box: tools/profile_round.sh r06 + pass sequences (two passes of the headline,
# config 4) + cryptomatte sequence, then the GPU suite as the driver runs it.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r06f/profiles
bash tools/profile_round.sh r06 > gpurun_out/r06f_profile_round.log 2>&1
cp gpurun_out/r06/profiles/* gpurun_out/r06f/profiles/ 2>/dev/null
PASSES=2 bash tools/pass_sequence.sh > gpurun_out/r06f/profiles/r06_pass_sequence.txt 2>&1
bash tools/pass_sequence.sh --lens petzval_58mm --aovs 8 > gpurun_out/r06f/profiles/r06_pass_sequence_config4.txt 2>&1
bash tools/crypto_sequence.sh > gpurun_out/r06f/profiles/r06_crypto_sequence.txt 2>&1
export GPU_MAX_HW_QUEUES=8
S=$(date +%s)
python3 -m pytest tests -m gpu -q -x > gpurun_out/r06f/profiles/r06_gpu_suite.log 2>&1; echo "rc=$? wall=$(( $(date +%s) - S )) s" >> gpurun_out/r06f/profiles/r06_gpu_suite.log
tail -3 gpurun_out/r06f/profiles/r06_gpu_suite.log
tail -c 600 gpurun_out/r06f/profiles/r06_bench_line.json
