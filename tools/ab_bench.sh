#!/bin/bash
# A/B of library builds on ONE box (boxes differ by ~5 %): headline step + optional extra bench flags, alternating runs.
# usage: tools/ab_bench.sh <tag> <libA> <libB> [... bench flags]
cd "$(dirname "$0")/.."
TAG=$1; A=$2; B=$3; shift 3
O=gpurun_out/ab_$TAG; mkdir -p $O
FLAGS="--steps 16 --warmup 4 --no-cpu-baseline --no-second-regime --no-configs --no-pcie --no-parity-check --no-scan-alone $@"
for rep in 1 2 3; do
  for L in $A $B; do
    n=$(basename $L .so)
    LENTIL_HIP_LIB=$PWD/$L timeout 600 python3 bench.py $FLAGS 2>>$O/err.log | tail -1 > $O/${n}_$rep.json
    python3 - <<PY
import json
d=json.load(open("$O/${n}_$rep.json")); print("%-28s rep $rep: %.4f ms  kernels %s  scan frac %.3f  rounds %s" % ("$n", d["ms_per_step"], d.get("kernels_ms"), d["roofline"]["frac"], d["passes"]["solve_accept_rounds_max"]))
PY
  done
done | tee $O/summary.txt
