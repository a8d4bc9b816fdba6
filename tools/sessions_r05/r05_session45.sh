#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s45; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
timeout 900 python3 -m pytest tests/test_crypto.py -m gpu -q -x > $O/pytest_crypto.log 2>&1; echo "rc=$?" >> $O/pytest_crypto.log
for rep in 1 2; do
  LENTIL_CRYPTO_LAZY_CLEAR=0 python3 tools/crypto_rate.py 2>/dev/null | tail -1 > $O/rate_eager_$rep.json
  python3 tools/crypto_rate.py 2>/dev/null | tail -1 > $O/rate_lazy_$rep.json
done
python3 - <<'PY' > $O/rates.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r05s45/rate_*.json")):
    d = json.loads(open(f).read())
    print(f.split("/")[-1], {k: (v.get("redistribute_ms"), v.get("clear_and_redistribute_ms"), v.get("added_ms_per_aov"), v.get("added_ms_per_aov_with_clear")) for k, v in d.items() if isinstance(v, dict)})
PY
bash tools/crypto_sequence.sh > $O/crypto_sequence.txt 2>&1
