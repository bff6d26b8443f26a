"""SURVEY.md 8(d)'s channel (tests/sig.py: Impair / SPEC) on the CPU: the resampler that makes the fractional delay and the clock error is
what it says it is, the default channel is untouched (the committed fixtures depend on it), the oracle's chains run on the impaired inputs,
and the clock-error cases of the GPU parity lists really drive the timing loops into their limiters (VERDICT r5 "next" #1: "a drift large
enough to hit max_dev is one of the cases")."""
import numpy as np
import pytest

import orc
import sig


def test_resampler_delays_and_compresses_a_tone():
    n = 20000
    f = 0.013   # cycles per sample
    x = np.exp(2j * np.pi * f * np.arange(n))
    y = sig.resample_clock(x, 0.37, 0.0)
    k = np.arange(100, n - 100)
    assert np.max(np.abs(y[k] - np.exp(2j * np.pi * f * (k - 0.37)))) < 1e-4   # Kaiser beta 8: ripple below -80 dB, far under every test noise floor
    y = sig.resample_clock(x, 0.37, 20.0)
    k = np.arange(100, y.size - 100)
    assert np.max(np.abs(y[k] - np.exp(2j * np.pi * f * (k - 0.37) * (1 + 20e-6)))) < 1e-4   # Kaiser beta 8: ripple below -80 dB, far under every test noise floor
    # +20 ppm: the transmitter's clock is fast, the waveform is compressed -- fewer samples come out
    assert y.size == int(np.floor((n - 1) / (1 + 20e-6) + 0.37))
    assert sig.resample_clock(x, 0.0, 0.0) is not None and np.array_equal(sig.resample_clock(x, 0.0, 0.0), x)


def test_default_channel_is_unchanged():
    """impair=None must reproduce the stream the committed fixtures were minted from (tests/golden/make_golden.py, seed 77)"""
    z = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "gmsk10k_1M.npz"))
    y, _ = sig.make_stream("gmsk10k", nframes=2, device_rate=1000000, rx_offset_hz=0.0, seed=77, amp=0.25)
    y = y[: y.size & ~1]
    assert np.array_equal(y.view(np.float32).astype(np.float16), z["iq_f16"])


def test_spec_noise_level_is_es_over_n0():
    rng = np.random.default_rng(1)
    x = np.exp(2j * np.pi * 0.01 * np.arange(200000))
    y = sig.channel(x, 1e6, 0.0, 99.0, 1.0, rng, impair=sig.Impair(0.0, 0.0, 12.0), samples_per_symbol=500.0)
    noise = y.astype(np.complex128) - x
    n0 = np.mean(np.abs(noise) ** 2)           # complex noise variance per sample
    assert abs(10 * np.log10(1.0 * 500.0 / n0) - 12.0) < 0.1


DEMODS = {
    "qpsk250k": (orc.demod_qpsk, dict(sps=2, filter_width=160000)), "qpsk20k": (orc.demod_qpsk, dict(sps=25, filter_width=6500)),
    "2fsk1k": (orc.demod_2fsk, dict(sps=10, filter_width=2000, fm=False)), "gmsk10k": (orc.demod_gmsk, dict(sps=1, filter_width=20000)),
    "4fsk2kfm": (orc.demod_4fsk, dict(sps=5, filter_width=3000, fm=True)), "4fsk100k": (orc.demod_4fsk, dict(sps=2, filter_width=125000, fm=True)),
    "4fsk2k": (orc.demod_4fsk, dict(sps=5, filter_width=4000, fm=False)), "bpsk1k": (orc.demod_bpsk, dict(sps=10)), "bpsk2k": (orc.demod_bpsk, dict(sps=5)),
}


@pytest.mark.parametrize("mode", sorted(sig.CLAMP_CASES))
def test_clock_error_cases_drive_the_loops_into_their_limiters(mode):
    """every row of sig.CLAMP_CASES (the inputs of test_gpu_parity.py::test_chain_bit_exact_clock_error_past_the_loop_clamp): the oracle's timing
    loop sits in its limiter at least `min_hits` times on one stream, and the number of symbols differs from the nominal one (stuff / skip)"""
    demod, kw = DEMODS[mode]
    impair, nframes = sig.clamp_impair(mode)
    y, _ = sig.make_stream(mode, nframes=nframes, seed=8, impair=impair)
    orc.loop_clamp_hits()
    r = demod(orc.frontend(y, 1000000, 0.0), **kw)
    hits = orc.loop_clamp_hits()
    assert r["bits_a"].size > 0
    assert hits >= sig.CLAMP_CASES[mode][2], (mode, impair, hits)


@pytest.mark.parametrize("mode,demod,kw", [
    ("gmsk10k", orc.demod_gmsk, dict(sps=1, filter_width=20000)),
    ("qpsk250k", orc.demod_qpsk, dict(sps=2, filter_width=160000)),
    ("4fsk100k", orc.demod_4fsk, dict(sps=2, filter_width=125000, fm=True)),
])
def test_survey_channel_moves_the_decisions(mode, demod, kw):
    """the 8(d) channel is a DIFFERENT test input, not a relabelled one: against the default channel with the same seed the received
    stream has another length and the chain's float port differs"""
    a, _ = sig.make_stream(mode, nframes=2, seed=5)
    b, _ = sig.make_stream(mode, nframes=2, seed=5, impair=sig.SPEC)
    assert a.size != b.size
    ra, rb = demod(orc.frontend(a, 1000000, 0.0), **kw), demod(orc.frontend(b, 1000000, 0.0), **kw)
    n = min(ra["filtered"].size, rb["filtered"].size)
    assert n > 100 and not np.array_equal(ra["filtered"][:n], rb["filtered"][:n])


@pytest.mark.parametrize("ppm,expect_hits", [(20.0, False), (-14000.0, True), (14000.0, True)])
def test_4fsk_symbol_clock_error_reaches_the_dmr_tails_limiter(ppm, expect_hits):
    """the inputs of test_gpu_chan.py::test_channelizer_64_4fsk_channels_with_symbol_clock_error: at 20 ppm the DMR tail's symbol_sync_ff (max_dev 0.06 of 5
    samples) tracks without touching its limiter, at -14000 / +14000 ppm it sits in it (orc.loop_clamp_hits), and the symbol count moves with the clock"""
    x, _ = sig.make_4fsk(nsym=600, seed=50, amp=0.3, noise=0.0, fs=1000000.0, clock_ppm=ppm, frac_delay=0.37)
    orc.loop_clamp_hits()
    r = orc.demod_dmr(x)
    hits = orc.loop_clamp_hits()
    assert r["bits_a"].size > 800
    assert (hits > 50) == expect_hits, (ppm, hits)
