#!/bin/bash
# usage: tools/bench_sweep.sh "VAR=a,b,c" ["VAR2=x,y"]  -- runs bench.py (short, no CPU baseline) for the cross product
# of the given environment settings and prints ms/step and the kernel split for each.
B="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-second-regime ${BENCH_ARGS}"
run() {
  echo -n "$* -> "
  env "$@" $B 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernels_ms'], round(d['value']), 'roofline', d['roofline']['frac'])"
}
if [ $# -eq 0 ]; then run X=1; exit; fi
IFS=',' read -ra A <<< "${1#*=}"; N1="${1%%=*}"
if [ $# -eq 1 ]; then for a in "${A[@]}"; do run "$N1=$a"; done; exit; fi
IFS=',' read -ra Bv <<< "${2#*=}"; N2="${2%%=*}"
for a in "${A[@]}"; do for b in "${Bv[@]}"; do run "$N1=$a" "$N2=$b"; done; done
