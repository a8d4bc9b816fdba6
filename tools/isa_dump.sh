#!/bin/bash
# Device assembly of liblentil_hip.so (the flags of __graft_entry__.build, from the sources as they are) into /tmp/dis/dev.s -- what
# profiles/r04_pmc_solve.txt and tools/isa_blocks.py read:   bash tools/isa_dump.sh && python3 tools/isa_blocks.py /tmp/dis/dev.s solve_po_kernel 50
# (extra compiler flags, e.g. -DLENTIL_SOLVE_ATTR=..., are passed through)
cd "$(dirname "$0")/.."
mkdir -p /tmp/dis
python3 - "$@" <<'PY'
import sys
import __graft_entry__ as g
a = sys.argv[1:]
g.hip_variant("/tmp/dis/dev.s", defines=[f for f in a if f.startswith("-D")], extra=[f for f in a if not f.startswith("-D")], device_asm=True)
PY
