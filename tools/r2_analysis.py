#!/usr/bin/env python3
"""Which items of the headline frame need more than their first batch, and could that be known when they are published?
Runs the oracle on the highlight visits alone (CPU): per item the attempts it used, against its geometry."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import common, oracle_lib
from pota_amd import capi, workload

W, H, M = 3840, 2160, 9
seed = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0x5EED
p, model, table, keep = common.po_setup(W, H, samples_override=1024)
n = W * H * M
chunk = 1 << 23
sel = {k: [] for k in ("rgba", "pos_z", "raydir_time", "volume_ignore", "transmission")}
idx_all = []
thr = 2.0 ** -16
for v0 in range(0, n, chunk):
    v1 = min(n, v0 + chunk)
    cols = workload.generate(np, v0, v1, W, H, M, seed=seed, f_hi=thr, focus_dist=150.0, tan_half_fov=common.tan_half_fov(p))
    hi = cols["rgba"][:, 0] > 50.0          # the generator's highlights carry HIGHLIGHT_RADIANCE
    ii = np.nonzero(hi)[0]
    idx_all.append(ii + v0)
    for k in sel:
        sel[k].append(cols[k][ii])
idx = np.concatenate(idx_all)
cols = {k: np.ascontiguousarray(np.concatenate(v)) for k, v in sel.items()}
cols["extra"] = []
pix = idx // M
cols["pixel"] = ((pix % W).astype(np.uint32) | ((pix // W).astype(np.uint32) << 16)).astype(np.uint32)
print("highlight visits:", idx.size)
visits, keepv = capi.make_visits(cols, visits_per_pixel=0)
orc = oracle_lib.load()
ref = common.run_oracle(orc, p, table, visits)
log = ref.log()
c = ref.counters()
print("redistributed", c.redistributed_visits, "attempted", c.attempted_draws, "accepted", c.accepted_draws)
used = np.zeros(idx.size, np.int64); acc = np.zeros(idx.size, np.int64)
np.maximum.at(used, log[:, 0], (log[:, 1] & 0x3FFFFFFF) + 1)
np.add.at(acc, log[:, 0], 1)
used[acc < 1024] = 5120
px = (pix % W).astype(np.int64); py = (pix // W).astype(np.int64)
# footprint of the item's accepted draws
lp = log[:, 2].astype(np.int64); lx = lp % (W + 1); ly = lp // (W + 1)
rad = np.zeros(idx.size)
for i in range(idx.size):
    m = log[:, 0] == i
    if m.any():
        rad[i] = max(np.abs(lx[m] - px[i]).max(), np.abs(ly[m] - py[i]).max())
edge = np.minimum(np.minimum(px, W - 1 - px), np.minimum(py, H - 1 - py))
need = used > 1024 + 16
print("items needing a second batch (used > samples + 16):", int(need.sum()), "of", idx.size)
print("  of them with edge distance < footprint radius:", int((need & (edge < rad)).sum()))
print("items NOT needing one but with edge distance < 1.1 x radius:", int((~need & (edge < 1.1 * rad)).sum()))
order = np.argsort(-used)
print("worst items: used, accepted, px, py, edge, radius")
for i in order[:45]:
    print("  ", used[i], acc[i], px[i], py[i], edge[i], int(rad[i]), "depth %.1f" % -cols["pos_z"][i, 2])
np.save("/tmp/r2_items_%x.npy" % seed, np.stack([used, acc, px, py, edge, rad, -cols["pos_z"][:, 2]], axis=1))
