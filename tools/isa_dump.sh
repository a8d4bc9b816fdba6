#!/bin/bash
# Device assembly of liblentil_hip.so (same flags as __graft_entry__.build) into /tmp/dis/dev.s -- what profiles/r04_pmc_solve.txt
# and tools/isa_blocks.py read:   bash tools/isa_dump.sh && python3 tools/isa_blocks.py /tmp/dis/dev.s solve_po_kernel 50
# (extra compiler flags, e.g. -DLENTIL_SOLVE_ATTR=..., are passed through)
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/dis
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -munsafe-fp-atomics -I $R/include --offload-device-only -S $R/pota_amd/csrc/lentil_hip.hip -o /tmp/dis/dev.s "$@"
