#!/bin/bash
# the shader clock while the bench's steps run (tools/clock_trace.py): headline pipelined / every step waited for, config 4, and the
# highlight-heavy frame (the solve kernel alone for 100 ms)
cd "$(dirname "$0")/../.."
O=gpurun_out/r06s25; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
timeout 300 python3 tools/clock_trace.py --steps 12 > $O/clock_headline.txt 2>&1
timeout 300 python3 tools/clock_trace.py --steps 12 --sync-each > $O/clock_headline_sync.txt 2>&1
timeout 300 python3 tools/clock_trace.py --steps 8 --lens petzval_58mm --aovs 8 > $O/clock_config4.txt 2>&1
timeout 300 python3 tools/clock_trace.py --steps 3 --f-hi 1.6e-3 --period-us 100 > $O/clock_heavy.txt 2>&1
tail -5 $O/clock_headline.txt; tail -4 $O/clock_headline_sync.txt; tail -4 $O/clock_config4.txt; tail -4 $O/clock_heavy.txt
