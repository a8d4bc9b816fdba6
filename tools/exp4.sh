#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/tl
for f in "" "-DLENTIL_EXP_NOCOUNT" "-DLENTIL_EXP_NOATOMIC" "-DLENTIL_EXP_NOTOUCH" "-DLENTIL_EXP_NOCOUNT -DLENTIL_EXP_NOATOMIC -DLENTIL_EXP_NOTOUCH"; do
  LENTIL_TL_FLAGS="$f" python3 tools/timeline.py --build > /dev/null 2>&1
  python3 tools/timeline.py --passes 7 --out gpurun_out/tl/exp4.txt > /dev/null 2>&1
  echo "== flags: $f"; grep -E "accept<1>|accept<2>" gpurun_out/tl/exp4.txt | cut -c1-80; tail -1 gpurun_out/tl/exp4.txt | cut -c60-260
done
