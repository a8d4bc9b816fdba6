#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
B="bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-second-regime --no-configs --no-pcie --no-parity-check --no-scan-alone"
for cfg in "LENTIL_ACCEPT_WIDE=0 LENTIL_ACCEPT_BLOCKS=4" "LENTIL_ACCEPT_WIDE=1 LENTIL_ACCEPT_BLOCKS=4" "LENTIL_ACCEPT_WIDE=1 LENTIL_ACCEPT_BLOCKS=2" "LENTIL_ACCEPT_WIDE=1 LENTIL_ACCEPT_BLOCKS=6" "LENTIL_ACCEPT_WIDE=0 LENTIL_ACCEPT_BLOCKS=2"; do
  rm -rf /tmp/kt; export $cfg LENTIL_STREAM=0 LENTIL_CHUNKS=1
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $B > /dev/null 2>&1
  f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1)
  echo "== $cfg"; grep -E "accept_kernel|solve_po|solve_slow|scan_dma" $f | awk -F, '{printf "  %-60s calls %s avg %.1f us min %.1f max %.1f\n", substr($1,1,60), $2, $4/1e3, $6/1e3, $7/1e3}'
  unset LENTIL_ACCEPT_WIDE LENTIL_ACCEPT_BLOCKS
done
