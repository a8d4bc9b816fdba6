#!/bin/bash
# what clock and power does the chip hold under the solve kernels / under the streamed headline pass?
cd "$(dirname "$0")/.."
O=gpurun_out/clock_${1:-x}; mkdir -p $O
rocm-smi --showclocks --showpower --showmaxpower > $O/idle.txt 2>&1
( for i in $(seq 1 60); do rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|fclk|socclk|Power" | tr '\n' ' '; echo; sleep 0.15; done ) > $O/heavy_samples.txt 2>&1 &
SM=$!
timeout 300 python3 tools/solve_workload.py double_gauss_50mm 1.6e-3 40 > $O/heavy.log 2>&1
kill $SM 2>/dev/null; wait $SM 2>/dev/null
( for i in $(seq 1 50); do rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|fclk|socclk|Power" | tr '\n' ' '; echo; sleep 0.15; done ) > $O/headline_samples.txt 2>&1 &
SM=$!
timeout 300 python3 bench.py --steps 2000 --warmup 4 --no-cpu-baseline --no-second-regime --no-configs --no-pcie --no-parity-check --no-scan-alone 2>/dev/null | tail -1 > $O/headline.json
kill $SM 2>/dev/null; wait $SM 2>/dev/null
tail -3 $O/heavy.log | cut -c1-200
sed -n '5,12p;30,34p' $O/heavy_samples.txt | cut -c1-400
echo ---; sed -n '8,14p;30,34p' $O/headline_samples.txt | cut -c1-400
cat $O/idle.txt | head -40
