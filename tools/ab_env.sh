#!/bin/bash
# A/B of two environment settings on ONE box, alternating: tools/ab_env.sh "A=1" "A=0" [reps] [extra bench flags]
cd "$(dirname "$0")/.."
EA=$1; EB=$2; R=${3:-4}; shift 3
B="python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-second-regime --no-configs --no-pcie --no-parity-check --no-scan-alone $@"
for r in $(seq 1 $R); do
  for E in "$EA" "$EB"; do
    echo -n "$E -> "; env $E $B 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["kernels_ms"])'
  done
done
