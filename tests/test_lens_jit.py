"""Run-time lens specialisation (pota_amd/csrc/lentil_lens_jit.h): the emitter and the compilation, without a GPU.

The reference compiles every lens into the plugin (include/auto_generated_lens_includes/load_lt_sample_aperture.h:4-47,
src/lentil.h:1308); the library emits and compiles a lens's straight-line code at run time for any table that has no kernel
built in.  Here: the C++ emitter writes, for the two shipped lenses, exactly what tools/gen_lens_code.py wrote into
csrc/generated/ (whose kernels the GPU suite checks bit for bit against the interpreter and the oracle), and hiprtc compiles
the four solve kernels for a table that is NOT built in.  tests/test_gpu_lens_jit.py runs them.
"""
import os
import re

import pytest

import common
from pota_amd import capi, lens_io

GEN = os.path.join(common.ROOT, "pota_amd", "csrc", "generated")


def _coefficients(src):
    a = src.index("__device__ __constant__ double kCoef")
    b = src.index("};", a)
    return [float.fromhex(x) for x in re.findall(r"-?0x[0-9a-f.]+p[+-]\d+", src[a:b])]


def _functions(src, name):
    a = src.index("static __device__ __forceinline__ void eval_bw")
    b = src.index("static __device__ __forceinline__ double transmittance")
    c = src.index("};", b)
    return src[a:b].replace(name, "K"), src[b:c].replace(name, "K")


@pytest.mark.parametrize("lens", ["double_gauss_50mm", "petzval_58mm"])
def test_emitter_writes_what_the_generator_wrote(lens):
    table, keep = lens_io.make_lens_table(lens_io.load_lens_json(lens))
    rc, src, log, seconds, code_bytes = capi.lens_jit_compile(table, compile=False)
    assert rc == 0
    with open(os.path.join(GEN, "lens_%s.h" % lens)) as f:
        ref = f.read()
    assert _coefficients(src) == _coefficients(ref)           # the same doubles in the same order (base terms and c * e derivatives)
    assert _functions(src, "kCoef_rt") == _functions(ref, "kCoef_" + lens)       # ... and the same operations on them, text for text


def test_a_table_that_is_not_built_in_compiles():
    """anamorphic_petzval_58mm (cylindrical outer pupil; BASELINE config 4's "anamorphic"): no kernel of it is built into the
    library.  hiprtc cross-compiles for gfx950 without a GPU; a machine without libhiprtc skips."""
    table, keep = lens_io.make_lens_table(lens_io.load_lens_json("anamorphic_petzval_58mm"))
    rc, src, log, seconds, code_bytes = capi.lens_jit_compile(table, compile=True)
    if rc != 0 and "hiprtc is not available" in log:
        pytest.skip("no libhiprtc on this machine")
    assert rc == 0, log[:2000]
    assert code_bytes > 100000 and "struct Lens_rt" in src and seconds < 120
