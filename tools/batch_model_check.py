"""The first-batch model (lentil_hip_debug_batch_estimate) against what the oracle says the items of a frame needed
(tools/item_need.py -> tools/data/item_need_*.npz).  GPU box.  usage: python3 tools/batch_model_check.py file.npz [...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common
from pota_amd import capi
for f in sys.argv[1:]:
    d = np.load(f)
    lens = "petzval_58mm" if "petzval" in f else "double_gauss_50mm"
    W, H, S = int(d["xres"]) - 1, int(d["yres"]) - 1, int(d["samples"])
    p, model, table, keep = common.po_setup(W, H, lens=lens, samples_override=S)
    ctx = capi.Context(0)
    ctx.set_params(p); ctx.set_lens(table)
    pos, acc, last = d["pos"], d["accepted"], d["last_attempt"]
    retries = int(p.vignetting_retries)
    need = np.where(acc < S, 5 * S, last + 1)
    looked = np.minimum(need + retries, 5 * S + retries)
    est = ctx.debug_batch_estimate(pos[:, :3], S)
    batch = est[:, 3].astype(np.int64)
    short = batch < looked
    plain = S + retries + 16
    print("%s: items %d short %d | batch total %d, plain total %d, looked-at total %d | items above plain %d"
          % (os.path.basename(f), len(need), int(short.sum()), int(batch.sum()), plain * len(need), int(looked.sum()), int((batch > plain).sum())))
    true_q = np.where(acc < S, acc / (5.0 * S), S / np.maximum(1.0, last + 1.0))
    for i in np.argsort(true_q)[:45]:
        print("   item %4d px %4d py %4d z %6.1f true %.3f | strict %.3f q %.3f fail %.3f | need %5d batch %5d %s"
              % (i, d["px"][i], d["py"][i], pos[i, 2], true_q[i], est[i, 0], est[i, 1], est[i, 2], need[i], batch[i], "SHORT" if short[i] else ""))
    waste = batch - looked
    print("   most traces beyond what was looked at:")
    for i in np.argsort(-waste)[:14]:
        print("   item %4d px %4d py %4d z %6.1f true %.3f | strict %.3f q %.3f fail %.3f | need %5d batch %5d"
              % (i, d["px"][i], d["py"][i], pos[i, 2], true_q[i], est[i, 0], est[i, 1], est[i, 2], need[i], batch[i]))
    ctx.close()
