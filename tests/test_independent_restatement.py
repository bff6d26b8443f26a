"""The oracle's recursive blocks against a second restatement written separately from SURVEY.md Appendix A (tests/appendix_a.py:
double precision, textbook structure).  Agreement = same hard decisions, floats within the float32 / summation-order noise of a
converged loop.  (Neither side is GNU Radio: this catches transcription errors, it does not pin the oracle -- DESIGN.md section 2.)"""
import numpy as np
import pytest

import appendix_a as A
import orc
import sig


def _rel(a, b):
    return np.max(np.abs(a - b)) / max(np.sqrt(np.mean(np.abs(b) ** 2)), 1e-12)


def test_design_helpers_agree():
    lo, up = A.fll_taps(10.0, 0.1, 16)
    # the oracle keeps its taps inside orc_fll_band_edge; compare through the block's behaviour below, and the tables directly:
    assert np.max(np.abs(A.mmse_table().astype(np.float32).ravel() - orc.table("mmse", 129 * 8))) < 2e-6
    assert np.max(np.abs(A.tanh_lut().astype(np.float32) - orc.table("tanh", 256))) < 2e-6
    assert np.allclose(np.abs(lo), np.abs(up))


@pytest.mark.parametrize("sps,rolloff,ntaps,bw,cfo", [(10.0, 0.1, 16, 24 * np.pi / 100, 0.04), (5.0, 0.35, 32, 8 * np.pi / 100, -0.03)])
def test_fll_band_edge(sps, rolloff, ntaps, bw, cfo):
    """a band-limited random signal with a carrier offset: both loops must pull it the same way (same derotated stream)"""
    rng = np.random.default_rng(3)
    n = 6000
    sym = rng.choice([-1.0, 1.0], n // int(sps) + 2)
    base = np.repeat(sym, int(sps))[:n] * 0.5
    x = (base * np.exp(1j * cfo * np.arange(n)) + 0.01 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
    got = orc.fll_band_edge(x, sps, rolloff, ntaps, bw)
    want = A.fll_band_edge(x.astype(complex), sps, rolloff, ntaps, bw)
    # With the reference's loop bandwidths (24 pi / 100 !) the loop is far from contractive: float32 and float64 runs of the SAME
    # algorithm drift apart after a few hundred samples.  Sample-wise agreement is therefore required where rounding has not yet
    # been amplified (the first 120 samples: 1e-4 relative), behavioural agreement afterwards (both track the offset).
    assert _rel(got[:120 + ntaps], want[:120 + ntaps]) < 1e-3

    def mean_rotation(y):   # average phase advance per sample of the derotated stream relative to the clean baseband
        r = y[ntaps + 1:] * np.conj(base[1:n - ntaps]) * np.conj(y[ntaps:-1] * np.conj(base[:n - ntaps - 1]))
        return np.angle(np.sum(r[n // 2:]))
    if bw < 0.5:   # (at 24 pi / 100 the long-run behaviour of two precisions is not comparable: the loop wanders)
        assert abs(mean_rotation(got) - mean_rotation(want)) < 0.01


@pytest.mark.parametrize("order,use_snr,bw", [(4, True, np.pi / 400), (2, False, 2 * np.pi / 200)])
def test_costas_loop(order, use_snr, bw):
    rng = np.random.default_rng(4)
    n = 8000
    if order == 4:
        s = (rng.choice([-1.0, 1.0], n) + 1j * rng.choice([-1.0, 1.0], n)) / np.sqrt(2)
    else:
        s = rng.choice([-1.0, 1.0], n).astype(complex)
    x = (s * np.exp(1j * (0.3 + 0.002 * np.arange(n))) + 0.02 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
    got = orc.costas(x, bw, order, use_snr)
    want = A.costas(x.astype(complex), bw, order, use_snr)
    assert _rel(got, want) < 2e-3


def test_agc2():
    rng = np.random.default_rng(5)
    x = (0.2 * (rng.standard_normal(4000) + 1j * rng.standard_normal(4000))).astype(np.complex64)
    x[2000:] *= 4
    assert _rel(orc.agc2(x, 1.0, 0.1, 1.0, 1.0), A.agc2(x.astype(complex), 1.0, 0.1, 1.0, 1.0)) < 1e-4


def _shaped(rng, levels, sps, nsym, frac=0.37, ppm=2e-5):
    """linear-interpolated, smoothed symbol stream with a fractional delay and a small clock error"""
    sym = rng.choice(levels, nsym)
    t = (np.arange(int(nsym * sps * (1 - 2 * ppm)) - 8) * (1 + ppm) + frac) / sps
    k = np.floor(t).astype(int)
    a = t - k
    pulse = 0.5 - 0.5 * np.cos(np.pi * a)       # raised-cosine transition between neighbouring symbols
    return sym, (sym[k] * (1 - pulse) + sym[np.minimum(k + 1, nsym - 1)] * pulse)


@pytest.mark.parametrize("ted,sps,loop_bw,max_dev,const,levels", [
    (orc.TED_MOD_MM, 10.0, 2 * np.pi / 200, 0.1, orc.CONST_BPSK, [-1.0, 1.0]),          # gr_demod_2fsk.cpp:106-110
    (orc.TED_MOD_MM, 4.0, 2 * np.pi / 200, 0.05, orc.CONST_BPSK, [-1.0, 1.0]),          # gr_demod_gmsk.cpp:89-92
    (orc.TED_MM, 5.0, 2 * np.pi / 100, 0.06, orc.CONST_4LEVEL, [-1.5, -0.5, 0.5, 1.5]),   # gr_demod_dmr.cpp:70-71
])
def test_symbol_sync_ff(ted, sps, loop_bw, max_dev, const, levels):
    rng = np.random.default_rng(6)
    sym, x = _shaped(rng, levels, sps, 1500)
    x = (x + 0.01 * rng.standard_normal(x.size)).astype(np.float32)
    got = orc.symbol_sync_ff(x, ted, sps, loop_bw, 1.0, 0.2869, max_dev, const)
    want = A.symbol_sync(x.astype(float), "mod_mm" if ted == orc.TED_MOD_MM else "mm", sps, loop_bw, 1.0, 0.2869, max_dev,
                         {orc.CONST_BPSK: "bpsk", orc.CONST_4LEVEL: "4level"}[const], False).real
    assert got.size == want.size and got.size > 1400
    assert np.max(np.abs(got - want)) < 5e-3
    # decisions: nearest level of both equals the transmitted symbol sequence after lock (up to the loop's own delay)
    lv = np.array(levels)
    dg, dw = lv[np.argmin(np.abs(got[:, None] - lv), axis=1)], lv[np.argmin(np.abs(want[:, None] - lv), axis=1)]
    assert np.array_equal(dg[200:], dw[200:])


def test_symbol_sync_cc_dqpsk():
    rng = np.random.default_rng(7)
    _, xi = _shaped(rng, [-1.0, 1.0], 2.0, 3000)
    _, xq = _shaped(rng, [-1.0, 1.0], 2.0, 3000)
    x = ((xi + 1j * xq) / np.sqrt(2) + 0.01 * (rng.standard_normal(xi.size) + 1j * rng.standard_normal(xi.size))).astype(np.complex64)
    got = orc.symbol_sync_cc(x, orc.TED_MOD_MM, 2.0, 2 * np.pi / 25000, 1.0, 0.2869, 8e-4, orc.CONST_DQPSK)   # gr_demod_qpsk.cpp:105-109
    want = A.symbol_sync(x.astype(complex), "mod_mm", 2.0, 2 * np.pi / 25000, 1.0, 0.2869, 8e-4, "dqpsk", True)
    # (loop bandwidth 2 pi / 25000: practically a free-running timing NCO; float32 vs float64 mu drift apart slowly)
    # and a sample lands now and then on the other side of a rint(mu * 128) boundary (next interpolator row: ~1 % of amplitude)
    assert got.size == want.size and _rel(got, want) < 5e-2
    assert np.mean(np.sign(got.real) == np.sign(want.real)) > 0.999 and np.mean(np.sign(got.imag) == np.sign(want.imag)) > 0.999


def test_viterbi_and_descrambler():
    """Integer blocks.  The oracle restates VOLK's SPIRAL kernel (the one x86 dispatches to and docs/OPERATION.md:4 names as the
    working one); Appendix A.9 describes the GENERIC kernel.  The separately written spiral restatement must agree with the oracle
    bit for bit on noisy soft symbols; the generic one agrees exactly without noise and differs only through tie-breaking with
    noise (both are maximum-likelihood survivors), which is why the two kernels are interchangeable for hard-decision parity only
    on clean signals -- a known limit of the unpinned oracle, recorded in DESIGN.md section 2."""
    rng = np.random.default_rng(8)
    bits = rng.integers(0, 2, 80 * 12, dtype=np.uint8)
    coded = orc.cc_encode_k7(orc.scramble(bits))
    clean = np.where(coded > 0, 255, 0).astype(np.uint8)
    soft = np.clip(np.rint(128 + 100 * (2.0 * coded - 1) + 40 * rng.standard_normal(coded.size)), 0, 255).astype(np.uint8)
    got = orc.cc_decode_k7(soft)
    spiral = A.cc_decode_k7(soft, variant="spiral")
    assert got.size == spiral.size and got.size >= 80 * 10
    assert np.array_equal(got, spiral)
    assert np.array_equal(orc.cc_decode_k7(clean), A.cc_decode_k7(clean, variant="generic"))
    assert np.mean(got == A.cc_decode_k7(soft, variant="generic")) > 0.995
    assert np.array_equal(orc.descramble(got), A.descramble(spiral))
    # and it decodes: after the descrambler's 8-bit lag the payload is back
    d = orc.descramble(got)
    assert np.mean(d[8:8 + 800] == bits[:800]) > 0.999
