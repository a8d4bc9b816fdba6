#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s25; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
timeout 900 python3 -m pytest tests/test_crypto.py -m gpu -q -x > $O/pytest_crypto.log 2>&1; echo "rc=$?" >> $O/pytest_crypto.log
for i in 1 2; do python3 tools/crypto_rate.py 2>/dev/null | tail -1 | python3 -c "
import sys, json
d=json.loads(sys.stdin.read()); print({k:(v.get('redistribute_ms'), v.get('added_ms_per_aov')) for k,v in d.items() if isinstance(v,dict)})"; done > $O/rate.txt 2>&1
bash tools/crypto_sequence.sh > $O/sequence.txt 2>&1
