#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s22; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
timeout 900 python3 -m pytest tests/test_crypto.py -m gpu -q -x > $O/pytest_crypto.log 2>&1; echo "rc=$?" >> $O/pytest_crypto.log
python3 tools/crypto_rate.py > $O/crypto_rate.json 2> $O/crypto_rate.err
python3 tools/crypto_rate.py > $O/crypto_rate2.json 2>> $O/crypto_rate.err
