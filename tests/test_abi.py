"""The C-ABI library loads and exports every symbol include/lentil_hip.h declares (no GPU needed);
the product path fails loudly -- never falls back -- when no GPU is present."""
import ctypes as C
import os
import re

import pytest

import common
from pota_amd import _abi, capi


def _declared(header):
    txt = open(os.path.join(common.ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lentil_(?:hip|host)_\w+)\s*\(", txt)))


def test_header_symbols_are_exported():
    lib = capi.load_library()
    names = _declared("lentil_hip.h")
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "liblentil_hip.so does not export %s" % n
    assert sorted(capi.EXPORTS) == names, "capi.EXPORTS out of sync with include/lentil_hip.h"
    assert lib.lentil_hip_abi_version() == 1


def test_host_library_exports():
    from pota_amd import bokeh
    lib = bokeh.load_host_library()
    for n in _declared("lentil_host.h"):
        assert hasattr(lib, n)


def test_struct_layouts_match_the_header():
    """sizeof of the ctypes mirrors vs the C structs (compiled here with gcc)."""
    import subprocess
    import tempfile
    src = r'''
#include <stdio.h>
#include "lentil_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(lentil_params), sizeof(lentil_term), sizeof(lentil_poly),
         sizeof(lentil_lens_table), sizeof(lentil_bokeh_table), sizeof(lentil_visits), sizeof(lentil_counters),
         sizeof(lentil_draw_record), sizeof(lentil_crypto_visits), sizeof(lentil_pass_totals));
  return 0;
}'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(common.ROOT, "include"), c, "-o", exe])
        sizes = [int(v) for v in subprocess.check_output([exe]).split()]
    mine = [C.sizeof(t) for t in (_abi.Params, _abi.Term, _abi.Poly, _abi.LensTable, _abi.BokehTable, _abi.Visits,
                                  _abi.Counters, _abi.DrawRecord, _abi.CryptoVisits, _abi.PassTotals)]
    assert sizes == mine


def test_no_gpu_is_a_loud_error():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = capi.load_library()
    h = C.c_void_p()
    rc = lib.lentil_hip_create(0, C.byref(h))
    assert rc == _abi.ERR_HIP and not h.value
    assert b"HIP" in lib.lentil_hip_last_error(None) or b"device" in lib.lentil_hip_last_error(None)
    with pytest.raises(capi.LentilError):
        capi.Context(0)


def test_product_path_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under pota_amd/ may reference it."""
    pkg = os.path.join(common.ROOT, "pota_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(root, f), errors="ignore").read()
                assert "liblentil_oracle" not in txt and "oracle_lib" not in txt and "oracle/" not in txt, f


def test_measuring_aids_build_and_load():
    """tools/micro/libclock_sampler.so (built by __graft_entry__.build(): the one-wave clock sampler bench.py runs beside a short loop for
    roofline.clock_in_pass) loads and exports what tools/clock_trace.py binds; bench.py knows the switch that leaves it out."""
    so = os.path.join(common.ROOT, "tools", "micro", "libclock_sampler.so")
    assert os.path.exists(so), "build() did not produce tools/micro/libclock_sampler.so"
    lib = C.CDLL(so)
    for n in ("sampler_start", "sampler_stamp", "sampler_read"):
        assert hasattr(lib, n), n
    import importlib.util
    spec = importlib.util.spec_from_file_location("clock_trace", os.path.join(common.ROOT, "tools", "clock_trace.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert callable(mod.clock_in_pass) and callable(mod.trace_steps)
    assert "--no-clock-trace" in open(os.path.join(common.ROOT, "bench.py")).read()
