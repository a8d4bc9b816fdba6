#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s24; mkdir -p $O
export GPU_MAX_HW_QUEUES=8 LENTIL_CRYPTO_RATE_ONLY=0,1,3
for v in "X=1" "LENTIL_AUX_PRIO=0" "LENTIL_AUX_PRIO=0 LENTIL_CRYPTO_TILE_BLOCKS=2" "LENTIL_AUX_PRIO=0 LENTIL_CRYPTO_TILE_BLOCKS=3" "LENTIL_AUX_PRIO=0 LENTIL_CRYPTO_TILE_BLOCKS=4" "LENTIL_CRYPTO_TILE_BLOCKS=3"; do
  echo -n "$v -> "
  env $v python3 tools/crypto_rate.py 2>/dev/null | tail -1 | python3 -c "
import sys, json
d=json.loads(sys.stdin.read()); print({k:(v.get('redistribute_ms'), v.get('added_ms_per_aov')) for k,v in d.items() if isinstance(v,dict)})"
done > $O/crypto_variants.txt 2>&1
