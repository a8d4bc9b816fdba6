"""Run-time lens specialisation on the GPU (pota_amd/csrc/lentil_lens_jit.h; round 5).

A lens table without a kernel built into the library (anamorphic_petzval_58mm: cylindrical outer pupil) is served by the table
interpreter until the kernel hiprtc compiles for it has arrived; both give the same frame, bit for bit -- and the oracle's.
"""
import numpy as np
import pytest

import common
from pota_amd import capi
from test_gpu_parity import check_frame, check_logs, gpu_run

pytestmark = pytest.mark.gpu


def test_interpreter_and_runtime_kernel_agree_bit_for_bit(orc, tmp_path, monkeypatch):
    monkeypatch.setenv("LENTIL_JIT_CACHE", str(tmp_path / "cache"))
    W, H, M = 96, 64, 9
    p, model, table, keep = common.po_setup(W, H, lens="anamorphic_petzval_58mm", samples_override=48)
    visits, cols = common.make_stream(p, W, H, M, f_hi=0.02)
    ref = common.run_oracle(orc, p, table, visits)
    assert ref.counters().accepted_draws > 2000
    ctx = capi.Context(0)
    try:
        # the table interpreter (lens mode 1 keeps the run-time kernel out, as it keeps a built-in one out)
        c = gpu_run(ctx, p, table, visits, lens_mode=1, compiled=False)
        assert not ctx.lens_is_compiled()
        log_i = common.sort_log(ctx.draw_log())
        acc_i, w_i = ctx.download_accum(0)
        check_logs(ctx, ref)
        check_frame(ctx, ref)
        st, _ = ctx.lens_jit_status()
        assert st in (1, 2), st                      # set_lens started the compilation (or found the table in this process already)
        ctx.lens_jit_wait(600.0)
        st, seconds = ctx.lens_jit_status()
        assert st == 2
        # the run-time kernel: three passes (chunked, then blind / streamed)
        for k in range(3):
            c2 = gpu_run(ctx, p, table, visits, lens_mode=0, compiled=False)
            assert (c2.redistributed_visits, c2.attempted_draws, c2.accepted_draws) == (c.redistributed_visits, c.attempted_draws, c.accepted_draws)
            assert np.array_equal(common.sort_log(ctx.draw_log()), log_i)      # every accepted (visit, attempt, pixel), bit for bit
            acc_j, w_j = ctx.download_accum(0)
            assert np.array_equal(w_j != 0, w_i != 0)
            check_frame(ctx, ref)
        # ... and once more through the interpreter, in the same place of the context's pass order as the last pass above:
        # the very same sums
        gpu_run(ctx, p, table, visits, lens_mode=1, compiled=False)
        acc_k, w_k = ctx.download_accum(0)
        assert np.array_equal(common.sort_log(ctx.draw_log()), log_i)
        check_frame(ctx, ref)
    finally:
        ctx.close()
    # another context of the process shares the entry (the cache file for the next process: the test below)
    ctx2 = capi.Context(0)
    try:
        ctx2.set_params(p); ctx2.set_lens(table)
        assert ctx2.lens_jit_status()[0] == 2
    finally:
        ctx2.close()


def test_the_cache_serves_the_next_process(tmp_path):
    """Two child processes with one cache directory: the first compiles, the second loads (no compile time reported)."""
    import json
    import subprocess
    import sys
    code = (
        "import sys, json, os; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import common\nfrom pota_amd import capi\n"
        "p, model, table, keep = common.po_setup(64, 48, lens='anamorphic_petzval_58mm', samples_override=16)\n"
        "ctx = capi.Context(0); ctx.set_params(p); ctx.set_lens(table); ctx.lens_jit_wait(600.0)\n"
        "print(json.dumps(ctx.lens_jit_status())); ctx.close()\n" % (common.ROOT, common.ROOT + "/tests"))
    import os
    env = dict(os.environ, LENTIL_JIT_CACHE=str(tmp_path / "c"))
    out = [json.loads(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900, check=True).stdout.strip().splitlines()[-1])
           for _ in range(2)]
    assert out[0][0] == 2 and out[0][1] > 1.0          # compiled: seconds
    assert out[1] == [2, 0.0]                          # from the cache
