#!/bin/bash
# streamed passes by the thousand on the last commit (defaults): the headline, config 2, config 4, an N = 8 band of config 5 -- stalls
# and redone passes counted by the library (passes.process in the bench line)
cd "$(dirname "$0")/../.."
O=gpurun_out/r06s43; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
P='import sys,json; d=json.loads(sys.stdin.read()); p=d["passes"]; print("%.4f ms" % d["ms_per_step"], "timed %s streamed %s redone %s rounds_max %s lost bets %s | process: %s" % (p["timed"], p["streamed"], p["chunks_redone_after_a_short_estimate"], p["solve_accept_rounds_max"], p["first_batch_model"]["lean_passes_lost"], p["process"]))'
F="--no-cpu-baseline --no-configs --no-pcie --no-second-regime --no-parity-check --no-scan-alone --no-clock-trace --warmup 3"
echo -n "headline, 3000 steps: "; timeout 600 python3 bench.py $F --steps 3000 2>/dev/null | tail -1 | python3 -c "$P"
echo -n "config 2, 4000 steps: "; timeout 600 python3 bench.py $F --steps 4000 --width 1920 --height 1080 --samples 256 2>/dev/null | tail -1 | python3 -c "$P"
echo -n "config 4, 800 steps: "; timeout 600 python3 bench.py $F --steps 800 --lens petzval_58mm --aovs 8 2>/dev/null | tail -1 | python3 -c "$P"
echo -n "config 5 band 8,0, 1500 steps: "; timeout 600 python3 bench.py $F --steps 1500 --width 7680 --height 4320 --samples 2048 --emulate 8,0 2>/dev/null | tail -1 | python3 -c "$P"
