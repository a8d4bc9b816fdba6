#!/bin/bash
# Single bands of the N-GPU frames, each run alone on ONE MI355X (bench.py --emulate N,r: rank r's band of the strong-scaled
# frame, pass only -- no exchange, even bands): what the multi-GPU step's per-rank pass will cost, for BASELINE.md section 4.
cd "$(dirname "$0")/.."
P='import sys,json; d=json.loads(sys.stdin.read()); print("%.3f ms" % d["ms_per_step"], d.get("kernels_ms"), "streamed %s rounds %s" % (d["passes"]["streamed"], d["passes"]["solve_accept_rounds_max"]))'
F="--no-cpu-baseline --no-configs --no-pcie --no-second-regime --no-parity-check --no-scan-alone --steps 8 --warmup 3"
echo "# headline frame 3840x2160, 1024 draws: whole frame, then bands"
echo -n "N=1      : "; timeout 300 python3 bench.py $F 2>/dev/null | tail -1 | python3 -c "$P"
for e in "2,0" "2,1" "4,0" "4,1" "8,0" "8,3" "8,7"; do echo -n "N,r=$e  : "; timeout 300 python3 bench.py $F --emulate $e 2>/dev/null | tail -1 | python3 -c "$P"; done
echo "# config 5, 7680x4320, 2048 draws"
C="$F --width 7680 --height 4320 --samples 2048"
echo -n "N=1      : "; timeout 300 python3 bench.py $C 2>/dev/null | tail -1 | python3 -c "$P"
for e in "2,0" "2,1" "4,0" "4,1" "8,0" "8,3" "8,7"; do echo -n "N,r=$e  : "; timeout 300 python3 bench.py $C --emulate $e 2>/dev/null | tail -1 | python3 -c "$P"; done
