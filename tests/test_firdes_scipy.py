"""Filter DESIGNS (coefficients, not tap counts) of the oracle AND of the product library against a third-party statement of
the same windowed-sinc method: scipy.signal.firwin / scipy.signal.windows (published, independent of this repository and of
the author's recollection of gr-filter/lib/firdes.cc).  GNU Radio's firdes::low_pass / band_pass_2 are the textbook designs --
ideal response x symmetric window, scaled to unit gain at DC / the band centre -- so the two must agree to float rounding.
The tap COUNT rule (firdes::compute_ntaps: A fs / (22 tw), made odd) stays [GR-MEM]; it is pinned separately by the literal tap
counts of SURVEY.md Appendix B (tests/test_oracle.py::test_low_pass_tap_counts).

Every call below is one the reference makes (file:line cited per case)."""
import ctypes as C

import numpy as np
import pytest
ss = pytest.importorskip("scipy.signal")   # boxes without scipy skip this file instead of failing collection

import orc
import qradiolink_amd as q

TOL = 1e-6

SCIPY_WIN = {orc.WIN_HAMMING: "hamming", orc.WIN_HANN: "hann", orc.WIN_BLACKMAN: "blackman", orc.WIN_RECT: "boxcar",
             orc.WIN_BH: "blackmanharris"}


def _product(fn, *args, complex_out=False, ntaps=None):
    lib = q.load_library()
    f = getattr(lib, fn)
    f.restype = C.c_int
    conv = [C.c_double(a) if isinstance(a, float) else C.c_int(a) for a in args]
    n = f(*conv, None) if ntaps is None else (ntaps | 1)
    buf = np.zeros(n * (2 if complex_out else 1), np.float32)
    assert f(*conv, buf.ctypes.data_as(C.c_void_p)) == n
    return buf.view(np.complex64) if complex_out else buf


@pytest.mark.parametrize("win", sorted(SCIPY_WIN))
@pytest.mark.parametrize("n", [17, 73, 419, 1045])
def test_windows_match_scipy(win, n):
    w = np.zeros(n, np.float32)
    assert orc.lib.orc_window(win, n, w.ctypes.data_as(C.c_void_p)) == n
    ref = ss.get_window(SCIPY_WIN[win], n, fftbins=False)
    assert np.max(np.abs(w - ref)) <= TOL


LOW_PASS = [
    # (gain, fs, fc, tw, window)                                   reference call site
    (1.0, 1e6, 10000.0, 10000.0, orc.WIN_BH),         # gr_demod_2fsk.cpp:82-84 (1:50, 419 taps) -- the C1 front end
    (1.0, 25e6, 480000.0, 100000.0, orc.WIN_BH),      # gr_demod_base.cpp:1333-1336 @ 25 Msps (1045 taps) -- C2
    (1.0, 100e6, 480000.0, 100000.0, orc.WIN_BH),     # the same @ 100 Msps (4181 taps) -- C3
    (2.0, 2e6, 40000.0, 40000.0, orc.WIN_BH),         # gr_demod_gmsk.cpp:80-83 (2/25)
    (1.0, 20000.0, 2000.0, 2000.0, orc.WIN_BH),       # gr_demod_2fsk.cpp:92 _filter
    (1.0, 20000.0, 2000.0, 2000.0, orc.WIN_HAMMING),               # gr_demod_2fsk.cpp:97 _symbol_filter
    (1.0, 80000.0, 20000.0, 20000.0, orc.WIN_HAMMING),             # gr_demod_gmsk.cpp:98
]


@pytest.mark.parametrize("gain,fs,fc,tw,win", LOW_PASS)
def test_low_pass_coefficients_match_scipy_firwin(gain, fs, fc, tw, win):
    got = orc.low_pass(gain, fs, fc, tw, win)
    ref = gain * ss.firwin(got.size, fc, window=SCIPY_WIN[win], fs=fs, scale=True)
    assert np.max(np.abs(got - ref)) <= TOL * max(1.0, gain)
    prod = _product("qrl_firdes_low_pass", gain, fs, fc, tw, win)
    assert prod.size == got.size and np.max(np.abs(prod - ref)) <= TOL * max(1.0, gain)


LOW_PASS_2 = [
    (1.0, 1e6, 250000.0, 50000.0, 60.0, orc.WIN_BH),      # gr_demod_qpsk.cpp:92-96 (1:2, 73 taps) -- C3 / C5
    (1.0, 1.6e6, 5000.0, 2000.0, 60.0, orc.WIN_BH),       # gr_demod_mmdvm_multi.cpp:62-66 at the C4 geometry (64 x 25 kHz)
    (1.0, 250000.0, 5000.0, 2000.0, 60.0, orc.WIN_BH),    # gr_demod_mmdvm_multi2.cpp:58-60 PFB prototype
    (3.0, 3e6, 5000.0, 2000.0, 60.0, orc.WIN_BH),         # gr_demod_dmr.cpp:55-58
]


@pytest.mark.parametrize("gain,fs,fc,tw,att,win", LOW_PASS_2)
def test_low_pass_2_coefficients_match_scipy_firwin(gain, fs, fc, tw, att, win):
    got = orc.low_pass_2(gain, fs, fc, tw, att, win)
    ref = gain * ss.firwin(got.size, fc, window=SCIPY_WIN[win], fs=fs, scale=True)
    assert np.max(np.abs(got - ref)) <= TOL * max(1.0, gain)
    prod = _product("qrl_firdes_low_pass_2", gain, fs, fc, tw, att, win)
    assert prod.size == got.size and np.max(np.abs(prod - ref)) <= TOL * max(1.0, gain)


def test_complex_band_pass_is_the_frequency_shifted_firwin_low_pass():
    # gr_demod_2fsk.cpp:93-96: complex_band_pass(1, 20000, -2000, 0, 2000, BH) / (1, 20000, 0, 2000, 2000, BH)
    for lo, hi in ((-2000.0, 0.0), (0.0, 2000.0)):
        got = orc.complex_band_pass(1.0, 20000.0, lo, hi, 2000.0, orc.WIN_BH)
        n = got.size
        lp = ss.firwin(n, (hi - lo) / 2, window="blackmanharris", fs=20000.0, scale=True)
        k = np.arange(n) - (n - 1) // 2
        ref = lp * np.exp(2j * np.pi * ((hi + lo) / 2) * k / 20000.0)
        assert np.max(np.abs(got - ref)) <= 2 * TOL   # the phase is accumulated in float like upstream: one more rounding per tap
        prod = _product("qrl_firdes_complex_band_pass", 1.0, 20000.0, lo, hi, 2000.0, orc.WIN_BH, complex_out=True)
        assert prod.size == n and np.max(np.abs(prod - ref)) <= 2 * TOL


def test_band_pass_2_matches_scipy_firwin_bandpass():
    # gr_demod_ssb.cpp:47-48: band_pass_2(1, 8000, 200, fw, 200, 90, BH)
    got = orc.band_pass_2(1.0, 8000.0, 200.0, 2700.0, 200.0, 90.0, orc.WIN_BH)
    ref = ss.firwin(got.size, [200.0, 2700.0], pass_zero=False, window="blackmanharris", fs=8000.0, scale=True)
    assert np.max(np.abs(got - ref)) <= TOL


def _rrc_textbook(gain, fs, symrate, alpha, ntaps):
    """Root raised cosine from the closed form in the literature (e.g. Proakis): with t in symbol periods,
    h(t) = [sin(pi t (1 - a)) + 4 a t cos(pi t (1 + a))] / [pi t (1 - (4 a t)^2)], h(0) = 1 - a + 4 a / pi,
    h(+-1/(4a)) = a / sqrt(2) [(1 + 2/pi) sin(pi/(4a)) + (1 - 2/pi) cos(pi/(4a))]; scaled so that sum(h) = gain."""
    ntaps |= 1
    t = (np.arange(ntaps) - ntaps // 2) * symrate / fs
    h = np.empty(ntaps)
    for i, x in enumerate(t):
        if x == 0:
            h[i] = 1 - alpha + 4 * alpha / np.pi
        elif abs(abs(4 * alpha * x) - 1) < 1e-9:
            h[i] = alpha / np.sqrt(2) * ((1 + 2 / np.pi) * np.sin(np.pi / (4 * alpha)) + (1 - 2 / np.pi) * np.cos(np.pi / (4 * alpha)))
        else:
            h[i] = (np.sin(np.pi * x * (1 - alpha)) + 4 * alpha * x * np.cos(np.pi * x * (1 + alpha))) / (np.pi * x * (1 - (4 * alpha * x) ** 2))
    return h * gain / h.sum()


@pytest.mark.parametrize("gain,fs,sr,alpha,ntaps", [
    (2.0, 2.0, 1.0, 0.35, 22),          # gr_demod_qpsk.cpp:100-103 (sps 2, 11 sps taps) -- C3 / C5
    (1.0, 24000.0, 4800.0, 0.2, 125),   # gr_demod_dmr.cpp:62-66 -- C4 4FSK tail
    (1.5, 20000.0, 4000.0, 0.2, 125),   # gr_demod_4fsk.cpp:130-133
    (4.0, 4.0, 1.0, 0.35, 44),          # a singular-point case: 4 alpha t = 1 is not on the grid here; alpha 0.25 below puts it on
    (1.0, 4.0, 1.0, 0.25, 33),          # t = 1 / (4 alpha) = 1 symbol is a grid point: the singular branch of firdes::root_raised_cosine
])
def test_root_raised_cosine_matches_the_textbook_closed_form(gain, fs, sr, alpha, ntaps):
    got = orc.root_raised_cosine(gain, fs, sr, alpha, ntaps)
    ref = _rrc_textbook(gain, fs, sr, alpha, ntaps)
    assert got.size == ref.size and np.max(np.abs(got - ref)) <= 2 * TOL * max(1.0, gain)
    prod = _product("qrl_firdes_root_raised_cosine", gain, fs, sr, alpha, ntaps, ntaps=ntaps)
    assert np.max(np.abs(prod - ref)) <= 2 * TOL * max(1.0, gain)


def test_gaussian_matches_scipy_window():
    # gr_mod_gmsk.cpp: firdes::gaussian(1, spb, bt 0.3, 4 spb): exp(-0.5 (t / sigma)^2), sigma = spb sqrt(ln 2) / (2 pi bt), unit sum
    spb, bt = 10.0, 0.3
    n = 40
    got = orc.gaussian(1.0, spb, bt, n)
    sigma = spb * np.sqrt(np.log(2.0)) / (2 * np.pi * bt)
    t = np.arange(n) - 0.5 * n + 1.0   # upstream's t0 = -0.5 ntaps, pre-incremented
    ref = np.exp(-0.5 * (t / sigma) ** 2)
    ref /= ref.sum()
    assert np.max(np.abs(got - ref)) <= TOL
