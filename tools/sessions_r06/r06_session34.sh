#!/bin/bash
# the bench line as the driver runs it, on the round's last commit (bench.py now carries roofline.clock_in_pass)
cd "$(dirname "$0")/../.."
O=gpurun_out/r06s34; mkdir -p $O
S=$(date +%s)
python3 bench.py --steps 20 --warmup 3 > $O/r06_bench_line_final_commit.json 2> $O/bench.err
echo "bench.py wall: $(( $(date +%s) - S )) s" > $O/wall.txt
cat $O/wall.txt; tail -c 2200 $O/r06_bench_line_final_commit.json
