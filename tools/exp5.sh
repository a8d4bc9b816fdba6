#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
B="bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-second-regime --no-configs --no-pcie --no-parity-check --no-scan-alone"
for cfg in "LENTIL_ACCEPT_WIDE=0" "LENTIL_ACCEPT_WIDE=1" "LENTIL_ACCEPT_WIDE=1 LENTIL_ACCEPT_BLOCKS=3" "LENTIL_ACCEPT_WIDE=1 LENTIL_ACCEPT_BLOCKS=5"; do
  rm -rf /tmp/kt
  env $cfg rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $B > /tmp/kt.log 2>&1
  f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1)
  echo "== $cfg  $(tail -1 /tmp/kt.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["kernels_ms"])' 2>/dev/null)"
  grep -E "accept_kernel|solve_slow|solve_po|scan_dma2|resolve" $f | awk -F, '{printf "  %-52s calls %s avg %.1f us min %.1f max %.1f\n", substr($1,1,52), $2, $4/1e3, $6/1e3, $7/1e3}'
done
