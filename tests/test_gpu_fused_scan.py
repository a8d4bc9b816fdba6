"""scan_solve_kernel (round 6): the scan of a streamed pass whose waves stay on the chip and go on as solve waves -- the third resident
solve wave per SIMD -- against the oracle, whole frames: accepted-draw lists bit for bit, counters, accumulators and the resolved image
(the pass resolves the frame behind the scan, which in this form no event marks: wait_scan_done_kernel).  Both scan bodies
(scan_dma2_kernel's for beauty-only frames, scan_dma_multi_kernel's for frames with gaussian AOV columns), both lenses compiled
into the library, and the plain form of the same frames beside them (LENTIL_FUSED_SCAN=0).
Reference loop: src/lentil_filter.cpp:248-299 (the draws), src/lentil.h:938-955 (the pixels' own sums the scan adds)."""
import pytest

from test_gpu_headline import _timed_config_vs_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lens", ["petzval_58mm", "double_gauss_50mm"])
@pytest.mark.parametrize("knob", ["default", "0"])
def test_frames_with_aov_columns(orc, monkeypatch, lens, knob):
    """BASELINE config 4's shape at 1280 x 720: eight gaussian AOVs beside the beauty (one scan block per CU, as the fused form needs),
    256 draws.  The default fuses such frames."""
    if knob != "default":
        monkeypatch.setenv("LENTIL_FUSED_SCAN", knob)
    _timed_config_vs_oracle(orc, "fused scan, %s, 8 AOV columns, LENTIL_FUSED_SCAN=%s" % (lens, knob), 1280, 720, lens, 256, n_extra=8,
                            kinds=[0] * 9, passes=(0, 1, 0), expect_form=(3, 1 if knob == "default" else 0))


@pytest.mark.parametrize("lens", ["double_gauss_50mm", "petzval_58mm"])
def test_beauty_only_frames(orc, monkeypatch, lens):
    """The headline's shape at 1920 x 1080 with 512 draws (enough for scan_cus_pct's blocks that do not scan): LENTIL_FUSED_SCAN=2."""
    monkeypatch.setenv("LENTIL_FUSED_SCAN", "2")
    _timed_config_vs_oracle(orc, "fused scan, %s, beauty only" % lens, 1920, 1080, lens, 512, passes=(0, 1, 0, 1), expect_form=(2, 1))
