"""CPU definition / known-answer tests of what round 4 added to the oracle (parity unpinned for the stock-block arithmetic, as everywhere:
these pin the restatement to its own definition and to behaviour any implementation of the block must show).
  * orc_chan_twiddles: the conjugate-symmetric DFT table of the channelizer contract;
  * orc_ctcss_squelch_ff: analog::ctcss_squelch_ff [GR-MEM] as gr_demod_nbfm::set_ctcss uses it (src/gr/gr_demod_nbfm.cpp:97-123);
  * orc_mod_am: gr_mod_am (src/gr/gr_mod_am.cpp:40-56)."""
import ctypes as C

import numpy as np

import orc


def _twiddles(M):
    W = np.zeros(M, np.complex64)
    orc.lib.orc_chan_twiddles(M, W.ctypes.data_as(C.c_void_p))
    return W


def test_channelizer_twiddles_are_exactly_conjugate_symmetric_and_correctly_rounded():
    for M in (2, 4, 10, 64):
        W = _twiddles(M)
        ideal = np.exp(2j * np.pi * np.arange(M) / M)
        assert np.abs(W.astype(np.complex128) - ideal).max() < 6e-8            # float rounding of the ideal table (sin(pi) -> 0 is 1.2e-16 away)
        for q in range(1, M):
            assert W[M - q].real.tobytes() == W[q].real.tobytes() and W[M - q].imag == -W[q].imag, (M, q)
        assert W[0] == 1 and (M % 2 or (W[M // 2].real == -1.0 and W[M // 2].imag == 0.0 and not np.signbit(W[M // 2].imag)))
        if M % 4 == 0:
            assert W[M // 4].imag == 1.0 and abs(W[M // 4].real) < 1e-7


def test_pfb_channelizer_bins_q_and_M_minus_q_are_mirror_sums():
    """what k_pfb_stream64 relies on: with the symmetric table the four chains of bin M - c are those of bin c with two signs flipped, so a
    REAL-valued branch vector gives exactly conjugate bins"""
    M = 64
    rng = np.random.default_rng(3)
    taps = np.zeros(M, np.float32); taps[:M] = 1.0                          # one tap per branch: v_p[n] = x[M n - p]
    x = np.zeros(M * 4, np.complex64); x.real = rng.standard_normal(x.size).astype(np.float32)
    out = np.zeros((M, x.size // M), np.complex64)
    orc.lib.orc_pfb_channelizer.restype = C.c_size_t
    n = orc.lib.orc_pfb_channelizer(x.ctypes.data_as(C.c_void_p), C.c_size_t(x.size), taps.ctypes.data_as(C.c_void_p), M, M, out.ctypes.data_as(C.c_void_p))
    assert n == x.size // M
    for c in range(1, M // 2):
        assert np.array_equal(out[M - c].real.view(np.uint32), out[c].real.view(np.uint32))
        assert np.array_equal((-out[M - c].imag + np.float32(0)).view(np.uint32), (out[c].imag + np.float32(0)).view(np.uint32))


def _tone(freq, n, amp=0.2, rate=8000, phase=0.0):
    return (amp * np.sin(2 * np.pi * freq * np.arange(n) / rate + phase)).astype(np.float32)


def test_ctcss_neighbours_follow_the_38_tone_table():
    fl, fr = C.c_float(), C.c_float()
    for tone, left, right in ((88.5, 85.4, 91.5), (67.0, 67.0 * 0.98, 71.9), (100.0, 97.4, 103.5)):
        orc.lib.orc_ctcss_freqs(C.c_float(tone), C.byref(fl), C.byref(fr))
        assert abs(fl.value - left) < 1e-4 and abs(fr.value - right) < 1e-4, (tone, fl.value, fr.value)
    orc.lib.orc_ctcss_freqs(C.c_float(125.0), C.byref(fl), C.byref(fr))   # not in the table: +-2 %
    assert abs(fl.value - 125.0 * 0.98) < 1e-3 and abs(fr.value - 125.0 * 1.02) < 1e-3


def test_ctcss_squelch_opens_on_its_tone_only():
    """gate = true: a muted squelch produces NO items.  Voice-band signal + the sub-audible tone -> output flows once the Goertzel block
    has seen the tone; the neighbouring tone of the table, a far tone, or no tone at all -> nothing comes out."""
    n = 8000 * 4
    voice = _tone(1000.0, n, 0.1)
    with_tone = orc.ctcss_squelch_ff(voice + _tone(88.5, n, 0.2), freq=88.5)
    assert with_tone.size > n // 2                                         # opened within the first Goertzel blocks
    tail = (voice + _tone(88.5, n, 0.2))[-with_tone.size // 2:]
    assert np.allclose(with_tone[-tail.size:], tail, atol=1e-6)            # fully open: the samples pass unchanged (unity ramp reached)
    for other in (85.4, 91.5, 250.3):
        assert orc.ctcss_squelch_ff(voice + _tone(other, n, 0.2), freq=88.5).size == 0, other
    assert orc.ctcss_squelch_ff(voice, freq=88.5).size == 0
    assert orc.ctcss_squelch_ff(voice + _tone(88.5, n, 0.004), freq=88.5).size == 0      # below level 0.01


def test_ctcss_squelch_ramps_and_closes_again():
    n = 8000
    x = np.concatenate([_tone(1000.0, 3 * n, 0.1) + _tone(88.5, 3 * n, 0.2), _tone(1000.0, 3 * n, 0.1)])
    y = orc.ctcss_squelch_ff(x, freq=88.5)
    assert 2 * n < y.size < 5 * n                                          # open for about the three seconds the tone lasts (+ detection delay, - release)
    k = np.flatnonzero(np.abs(y) > 1e-9)[0]
    first = np.abs(y[k:k + 400])
    assert first[:40].max() < first[200:400].max()                         # attack ramp of 160 samples: the first items are attenuated


def test_mod_am_carrier_and_modulation():
    """gr_mod_am: audio -> band-pass -> x bb_gain... -> + carrier 0.5 -> feed-forward AGC -> interpolator to 1 Msps -> band-pass.  Silence
    gives a constant-envelope carrier; a tone gives an envelope at the tone's frequency; output length = 125 samples per audio sample."""
    n = 1600
    quiet = orc.mod_am(np.zeros(n, np.float32))
    assert quiet.size == n * 125
    env = np.abs(quiet[quiet.size // 2:])
    assert env.mean() > 0.05 and env.std() / env.mean() < 0.02
    y = orc.mod_am(_tone(1000.0, n, 0.3))
    env = np.abs(y[y.size // 2:]).astype(np.float64)
    env -= env.mean()
    spec = np.abs(np.fft.rfft(env * np.hanning(env.size)))
    peak_hz = np.argmax(spec[5:]) + 5
    assert abs(peak_hz * 1e6 / env.size - 1000.0) < 30.0                   # the envelope carries the 1 kHz tone
    assert env.std() > 0.01
