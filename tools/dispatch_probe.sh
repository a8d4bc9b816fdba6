#!/bin/bash
# Development aid: which XCDs' shares of the first accept's grid are not dispatched while the second round's resident kernels
# wait beside it (LENTIL_OVERLAP_ACCEPT=1), and what is resident there.  Builds pota_amd/_ab/liblentil_hip_probe.so
# (-DLENTIL_PROBE_BUILD: per-XCD counters at the waves' entry and exit, a snapshot when the accept's last item is done with
# blocks of its grid not begun) and loops the tests that have a second round in flight.
#   tools/dispatch_probe.sh build        (here: hipcc cross-compiles)
#   tools/dispatch_probe.sh [runs]       (on the GPU box; prints the library's "[probe]" lines)
cd "$(dirname "$0")/.."
SO=pota_amd/_ab/liblentil_hip_probe.so
if [ "$1" = "build" ]; then
  mkdir -p pota_amd/_ab
  python3 - <<'PY'
import __graft_entry__ as g
g.hip_variant("pota_amd/_ab/liblentil_hip_probe.so", defines=["-DLENTIL_PROBE_BUILD"])
print("built pota_amd/_ab/liblentil_hip_probe.so")
PY
  exit $?
fi
export GPU_MAX_HW_QUEUES=8 LENTIL_HIP_LIB=$PWD/$SO LENTIL_OVERLAP_ACCEPT=1 LENTIL_DISPATCH_PROBE=1 LENTIL_STREAM_DEBUG=1
N=${1:-16}
for rep in $(seq 1 $N); do
  timeout 600 python3 -m pytest tests/test_gpu_batch_model.py -q -s 2>&1 | grep -h "\[probe\]\|\[stream\] note\|passed\|failed" | cut -c1-900
done
