"""The side outputs of gr_demod_base -- rssi_block (reference src/gr/rssi_block.cpp:31-44) and rx_fft_c's window / power spectrum
(src/gr/rx_fft.cpp:83-96,126-127) -- in the oracle: the deterministic log2 both sides share, the RSSI chain against an independent
float64 numpy model of the same blocks, the spectrum against numpy's FFT."""
import numpy as np

import orc


def test_det_log2f_is_log2_to_float_rounding():
    rng = np.random.default_rng(1)
    xs = np.concatenate([np.float32(2.0) ** rng.uniform(-140, 127, 4000).astype(np.float32), np.float32([1.0, 2.0, 0.5, 1.41421354, 1.4142137, 1e-45, 3e38])])
    got = np.array([orc.det_log2f(x) for x in xs], np.float32)
    want = np.log2(xs.astype(np.float64))
    ulp = np.spacing(np.abs(want).astype(np.float32)).astype(np.float64)
    assert np.max(np.abs(got.astype(np.float64) - want) / np.maximum(ulp, 1e-45)) <= 0.51
    # VOLK's log2f_non_ieee: infinities become +-127
    assert orc.det_log2f(0.0) == -127.0 and orc.det_log2f(float("inf")) == 127.0 and np.isnan(orc.det_log2f(-1.0))


def _rssi_model(x, level):
    """float64 model: moving SUM of 2000 powers (scale 1), y = 0.04 s + 0.96 y, 10 log10 y + level"""
    p = np.abs(x.astype(np.complex128)) ** 2
    c = np.concatenate([[0.0], np.cumsum(p)])
    idx = np.arange(1, p.size + 1)
    s = c[idx] - c[np.maximum(idx - 2000, 0)]
    y = np.zeros(p.size)
    prev = 0.0
    for i in range(p.size):
        prev = 0.04 * s[i] + 0.96 * prev
        y[i] = prev
    with np.errstate(divide="ignore"):
        return 10 * np.log10(y) + level


def test_rssi_block_matches_float64_model():
    rng = np.random.default_rng(7)
    n = 9000
    amp = np.where(np.arange(n) < 4000, 0.02, 0.3)       # a level step: window and IIR transients
    x = (amp * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
    got = orc.rssi_block(x, level=-12.5)
    want = _rssi_model(x, -12.5)
    assert np.max(np.abs(got[1:] - want[1:])) < 2e-4      # dB; float32 running sum against float64
    # steady state: 2000 x mean power, through an IIR of unit DC gain
    assert abs(got[-1] - (10 * np.log10(2000 * 2 * 0.3 ** 2) - 12.5)) < 0.3


def test_rssi_block_zero_input_reports_the_log_floor():
    got = orc.rssi_block(np.zeros(10, np.complex64), level=3.0)
    assert np.allclose(got, -127.0 / np.log2(10.0) * 10 + 3.0, atol=1e-3)   # log2f_non_ieee(0) = -127


def test_power_spectrum_matches_numpy_fft():
    rng = np.random.default_rng(3)
    n = 4096
    t = np.arange(n)
    x = (0.5 * np.exp(2j * np.pi * 300.25 * t / n) + 0.01 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
    w = np.blackman(n).astype(np.float32)
    got = orc.power_spectrum(x, w)
    X = np.fft.fftshift(np.fft.fft(x.astype(np.complex128) * w)) / n
    want = 10 * np.log10(np.abs(X) ** 2)
    assert np.max(np.abs(got - want)) < 1e-3
    assert np.argmax(got) == n // 2 + 300                 # the tone, negative frequencies first
