#!/bin/bash
cd "$(dirname "$0")/../.."
bash tools/profile_round.sh r05 > gpurun_out/r05_profile_round.log 2>&1
PASSES=1 bash tools/pass_sequence.sh > gpurun_out/r05/profiles/r05_pass_sequence.txt 2>&1
bash tools/pass_sequence.sh --lens petzval_58mm --aovs 8 > gpurun_out/r05/profiles/r05_pass_sequence_config4.txt 2>&1
bash tools/crypto_sequence.sh > gpurun_out/r05/profiles/r05_crypto_sequence.txt 2>&1
