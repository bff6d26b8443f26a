"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.
Bit-exact for hard-decision bits AND for the float side outputs (the kernels follow the oracle's
arithmetic contract, oracle/orc.h)."""
import numpy as np
import pytest

import orc
import sig

pytestmark = pytest.mark.gpu


def _run(qrl_ctx, mode_name, modem, device_rate, offset, B, chunk, nframes=3, seed=3, options=(), impair=None):
    import torch
    import qradiolink_amd as q
    iq = sig.make_batch(mode_name, B, nframes=nframes, device_rate=device_rate, rx_offset_hz=offset, seed=seed, impair=impair)
    dem = q.Demod(qrl_ctx, modem, batch=B, max_chunk=chunk, device_samp_rate=device_rate, carrier_offset_hz=offset)
    for opt, val in options:
        dem.set_option(opt, val)
    out = q.collect(dem, torch.from_numpy(iq).cuda(), chunk)
    dem.close()
    return iq, out


def _oracle(mode_name, x, device_rate, offset):
    fe = orc.frontend(x, device_rate, offset)
    if mode_name.startswith("2fsk"):
        return orc.demod_2fsk(fe, sps=10, filter_width=2500 if mode_name.endswith("fm") else 2000, fm=mode_name.endswith("fm"))
    if mode_name == "gmsk10k":
        return orc.demod_gmsk(fe, sps=1, filter_width=20000)
    if mode_name == "gmsk1k":
        return orc.demod_gmsk(fe, sps=10, filter_width=2000)
    if mode_name == "qpsk250k":
        return orc.demod_qpsk(fe, sps=2, filter_width=160000)
    if mode_name == "qpsk2k":
        return orc.demod_qpsk(fe, sps=125, filter_width=1300)
    if mode_name == "qpsk20k":
        return orc.demod_qpsk(fe, sps=25, filter_width=6500)
    if mode_name == "4fsk2k":
        return orc.demod_4fsk(fe, sps=5, filter_width=4000, fm=False)
    if mode_name.startswith("4fsk"):
        sps, fw = {"4fsk2kfm": (5, 3000), "4fsk1kfm": (10, 2000), "4fsk10kfm": (1, 20000), "4fsk100k": (2, 125000)}[mode_name]
        return orc.demod_4fsk(fe, sps=sps, filter_width=fw, fm=True)
    if mode_name.startswith("bpsk"):
        return orc.demod_bpsk(fe, sps=10 if mode_name == "bpsk1k" else 5)
    raise ValueError(mode_name)


def _compare(iq, out, mode_name, device_rate, offset):
    for b in range(iq.shape[0]):
        ref = _oracle(mode_name, iq[b], device_rate, offset)
        for port in ("bits_a", "bits_b"):
            if (mode_name.startswith("qpsk") or mode_name.startswith("4fsk")) and port == "bits_b":
                continue   # single-branch mode (gr_demod_qpsk.cpp:124-126): port 2 only
            assert out[port][b].size == ref[port].size, (port, b, out[port][b].size, ref[port].size)
            assert np.array_equal(out[port][b], ref[port]), "%s stream %d differs" % (port, b)
        for port in ("filtered", "constellation"):
            # bit-identical up to the sign of an exact zero (x + 0.0 maps -0 to +0): products that underflow below the
            # smallest denormal at the very first samples of a stream may come out as -0 on one side and +0 on the other
            got, want = out[port][b].view(np.float32) + np.float32(0), ref[port].view(np.float32) + np.float32(0)
            assert got.size == want.size, (port, b, got.size, want.size)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "%s stream %d not bit-identical" % (port, b)
        assert ref["bits_a"].size > 0


# the parity list: every chain family behind every front-end geometry (mode, modem type, device rate, samples per call)
PARITY_LIST = [
    ("gmsk10k", 22, 1000000, 1 << 20),
    ("gmsk10k", 22, 4000000, 1 << 22),
    ("2fsk1k", 18, 1000000, 1 << 21),
    ("2fsk1kfm", 16, 1000000, 1 << 21),
    ("gmsk1k", 21, 2000000, 1 << 22),
    ("gmsk10k", 22, 25000000, 1 << 23),     # front end 25:1, 1045 taps: k_decim_pm with three lag tiles (rounds 1-2: banded-Toeplitz MFMA decimator)
    ("gmsk10k", 22, 10000000, 1 << 23),     # front end 10:1, 419 taps
    ("qpsk250k", 26, 1000000, 1 << 20),     # C3 chain at the internal rate: agc2, 2x Costas, symbol_sync_cc, diff_phasor
    ("qpsk250k", 26, 10000000, 1 << 23),    # C3 behind the 10:1 front end
    ("gmsk10k", 22, 64000000, 1 << 24),     # front end 64:1, 2677 taps: 8-block tile, 22 loads per thread (NLD = 36 variant)
    ("gmsk10k", 22, 100000000, 1 << 24),    # front end 100:1, 4181 taps: only the 4-block MFMA tile fits the LDS
    ("qpsk250k", 26, 100000000, 1 << 24),   # BASELINE config C3 literally: QPSK-250k behind the 100:1 front end
    ("qpsk2k", 7, 1000000, 1 << 21),        # gr_demod_qpsk sps >= 125: 1:100 (3621 taps), FLL(32 taps), 5 samples per symbol
    ("qpsk20k", 1, 2000000, 1 << 22),       # gr_demod_qpsk 4 < sps < 125: 1:25, FLL, 4 samples per symbol
    ("4fsk2k", 3, 1000000, 1 << 21),        # gr_demod_4fsk non-FM branch: 4 band-pass magnitudes -> discriminator -> 837-tap LPF -> symbol_sync_cc
    ("4fsk2kfm", 5, 1000000, 1 << 21),      # gr_demod_4fsk FM branch: 4-level symbol_sync_ff, phase_modulator, (imag, real) soft pairs
    ("4fsk1kfm", 6, 2000000, 1 << 22),
    ("4fsk10kfm", 4, 4000000, 1 << 22),     # 2/25 resampler to 80 ksps, 8 samples per symbol
    ("4fsk100k", 27, 1000000, 1 << 20),     # 1:2 decimation to 500 ksps, 5 samples per symbol
    ("bpsk1k", 24, 1000000, 1 << 21),       # gr_demod_bpsk: FLL(32 taps) -> RRC -> agc2 -> clock_recovery_mm_cc -> costas(2)
    ("bpsk2k", 0, 2000000, 1 << 22),
]


@pytest.mark.parametrize("mode_name,modem,rate,chunk", PARITY_LIST)
def test_chain_bit_exact_single_call(qrl_ctx, mode_name, modem, rate, chunk):
    offset = 25000.0 if rate >= 2000000 else 1200.0
    iq, out = _run(qrl_ctx, mode_name, modem, rate, offset, B=3, chunk=chunk, nframes=2)
    _compare(iq, out, mode_name, rate, offset)


@pytest.mark.parametrize("mode_name,modem,rate,chunk", PARITY_LIST)
def test_chain_bit_exact_survey_8d_channel(qrl_ctx, mode_name, modem, rate, chunk):
    """The whole parity list behind SURVEY.md 8(d)'s channel (sig.SPEC: fractional delay 0.37 sample, clock error +20 ppm, Es/N0 12 dB per
    channel symbol) -- VERDICT r5 "missing" #3: a sliding symbol phase and marginal decisions through symbol_sync_ff (gr_demod_2fsk.cpp:106-110),
    symbol_sync_cc (gr_demod_qpsk.cpp:105-109) and clock_recovery_mm_cc (gr_demod_bpsk.cpp:51-103).  One call per stream."""
    offset = 25000.0 if rate >= 2000000 else 1200.0
    iq, out = _run(qrl_ctx, mode_name, modem, rate, offset, B=3, chunk=chunk, nframes=2, seed=5, impair=sig.SPEC)
    _compare(iq, out, mode_name, rate, offset)


# call cuts under the drifting clock: the recursions carry mu / omega / the interpolator window across calls while the symbol phase slides
@pytest.mark.parametrize("mode_name,modem,rate,chunk", [
    ("2fsk1k", 18, 1000000, 50000), ("2fsk1kfm", 16, 1000000, 33334), ("gmsk10k", 22, 1000000, 30000), ("gmsk10k", 22, 25000000, 750000),
    ("gmsk1k", 21, 2000000, 100002), ("qpsk250k", 26, 1000000, 33334), ("qpsk250k", 26, 100000000, 3000000), ("qpsk2k", 7, 1000000, 60000),
    ("qpsk20k", 1, 2000000, 100002), ("4fsk2k", 3, 1000000, 50000), ("4fsk2kfm", 5, 1000000, 50000), ("4fsk10kfm", 4, 4000000, 200000),
    ("4fsk100k", 27, 1000000, 30000), ("bpsk1k", 24, 1000000, 65536), ("bpsk2k", 0, 2000000, 100002),
])
def test_chain_bit_exact_survey_8d_channel_cut_into_calls(qrl_ctx, mode_name, modem, rate, chunk):
    offset = 25000.0 if rate >= 2000000 else 1200.0
    iq, out = _run(qrl_ctx, mode_name, modem, rate, offset, B=2, chunk=chunk, nframes=2, seed=6, impair=sig.SPEC)
    _compare(iq, out, mode_name, rate, offset)


# clock errors past the loops' clamps (sig.CLAMP_CASES, each measured with the oracle's limiter-hit counter in tests/test_channel_8d.py):
# symbol_sync_cc max_dev 8e-4 of 2 samples per symbol (QPSK-250k: 400 ppm; the narrow loop needs 1e5 symbols to get there, hence 12 frames),
# symbol_sync_ff max_dev 0.1 sample of 10 (1 %), clock_recovery_mm_cc's omega limit 1e-3 relative (30 frames); the loop sits in its limiter
# and the interpolator slips symbols -- same bits, same floats as the oracle, in one call and cut into calls
@pytest.mark.parametrize("mode_name,modem,rate,chunk", [
    ("qpsk250k", 26, 1000000, 1 << 20), ("qpsk250k", 26, 1000000, 33334), ("qpsk250k", 26, 10000000, 333340),
    ("qpsk20k", 1, 1000000, 60000), ("4fsk2k", 3, 1000000, 50000),
    ("2fsk1k", 18, 1000000, 1 << 21), ("2fsk1k", 18, 1000000, 50000),
    ("gmsk10k", 22, 1000000, 30000), ("gmsk10k", 22, 25000000, 750000),
    ("4fsk2kfm", 5, 1000000, 50000), ("4fsk100k", 27, 1000000, 30000),
    ("bpsk1k", 24, 1000000, 1 << 21), ("bpsk2k", 0, 1000000, 200000),
])
def test_chain_bit_exact_clock_error_past_the_loop_clamp(qrl_ctx, mode_name, modem, rate, chunk):
    offset = 25000.0 if rate >= 2000000 else 1200.0
    impair, nframes = sig.clamp_impair(mode_name)
    iq, out = _run(qrl_ctx, mode_name, modem, rate, offset, B=2, chunk=chunk, nframes=nframes, seed=8, impair=impair)
    _compare(iq, out, mode_name, rate, offset)
@pytest.mark.parametrize("mode_name,modem,chunk", [("2fsk1k", 18, 1 << 21), ("2fsk1k", 18, 50000), ("bpsk1k", 24, 1 << 21), ("bpsk1k", 24, 65536),
                                                   ("qpsk2k", 7, 60000)])
def test_fll_slim_geometry_bit_exact(qrl_ctx, mode_name, modem, chunk):
    """QRL_OPT_FLL_SLIM = 1 (k_fll<NT, 64, 16>: single-wave FLL workgroups with 16-sample windows) is declared result-neutral in
    include/qrl_hip.h: the chains that contain an FLL, in one call and cut into calls, against the oracle with the option on."""
    import qradiolink_amd as q
    iq, out = _run(qrl_ctx, mode_name, modem, 1000000, 1200.0, B=3, chunk=chunk, nframes=2, options=[(q.OPT_FLL_SLIM, 1)])
    _compare(iq, out, mode_name, 1000000, 1200.0)


@pytest.mark.parametrize("mode_name,modem,rate,chunk", [("2fsk1k", 18, 1000000, 1 << 21), ("2fsk1k", 18, 1000000, 50000), ("gmsk10k", 22, 4000000, 1 << 22),
                                                        ("gmsk10k", 22, 4000000, 100002), ("qpsk250k", 26, 1000000, 33334), ("qpsk250k", 26, 25000000, 1 << 23)])
def test_time_domain_scope_tap_bit_exact(qrl_ctx, mode_name, modem, rate, chunk):
    """gr_demod_base::enable_time_domain (src/gr/gr_demod_base.cpp:62-63, 1115-1147): _demod_valve -> rational_resampler_ccf(1, 10,
    low_pass(1, 1e6, 50000, 25000, HAMMING)) -> gr_sample_sink.  The 100 ksps scope items of every call equal the oracle's decimator on
    the oracle's 1 Msps front-end signal bit for bit, whether the tap reads the caller's rotated IQ (1 Msps device) or the front-end
    ring (4 / 25 Msps), in one call and cut into calls; the demodulator's own outputs are unchanged by the tap."""
    import torch
    import qradiolink_amd as q
    offset = 25000.0 if rate >= 2000000 else 1200.0
    iq = sig.make_batch(mode_name, 2, nframes=1, device_rate=rate, rx_offset_hz=offset, seed=41)
    n = iq.shape[1] & ~1
    dem = q.Demod(qrl_ctx, modem, batch=2, max_chunk=min(chunk, n), device_samp_rate=rate, carrier_offset_hz=offset)
    dem.enable_time_domain()
    d = torch.from_numpy(iq).cuda()
    parts, bits = [[], []], [[], []]
    for s0 in range(0, n, chunk):
        part = d[:, s0:min(s0 + chunk, n)]
        if part.shape[1] & 1:
            part = part[:, :-1]
        out = dem.process(part.contiguous())
        cnt, sc = dem.scope_counts.cpu().numpy(), dem.scope.cpu().numpy()
        c4 = out["counts"].cpu().numpy()
        for b in range(2):
            parts[b].append(sc[b, :cnt[b]].copy())
            bits[b].append(out["bits_a"][b, :c4[b, 2]].cpu().numpy().copy())
    dem.close()
    used = sum((min(chunk, n - s0) & ~1) for s0 in range(0, n, chunk))
    taps = orc.low_pass(1, 1000000, 50000, 25000)
    for b in range(2):
        fe = orc.frontend(iq[b, :used], rate, offset)
        want = orc.decim_auto(fe, taps, 10).view(np.float32) + np.float32(0)
        got = np.concatenate(parts[b]).view(np.float32) + np.float32(0)
        assert got.size == want.size and got.size > 1000, (got.size, want.size)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "stream %d" % b
        assert np.array_equal(np.concatenate(bits[b]), _oracle(mode_name, iq[b, :used], rate, offset)["bits_a"])


@pytest.mark.parametrize("rate,sr,fw", [(1000000, 50000, 0.0), (1000000, 200000, 0.0), (1000000, 0, 30000.0), (4000000, 50000, 10000.0), (1000000, 8000, 0.0)])
def test_time_domain_scope_tap_rate_and_filter_width(qrl_ctx, rate, sr, fw):
    """gr_demod_base::set_time_sink_samp_rate / set_time_domain_filter_width (src/gr/gr_demod_base.cpp:1249-1301) as qrl_demod_config.time_domain_samp_rate /
    time_domain_filter_width: decimation 1e6 / samp_rate with low_pass(1, 1e6, sr / 2 - sr / 8, sr / 4, HAMMING) (integer divisions), or low_pass(1, 1e6, w, w,
    HAMMING) after set_time_domain_filter_width -- bit-exact against the oracle's decimator on the oracle's 1 Msps front-end signal, cut into calls"""
    import torch
    import qradiolink_amd as q
    offset = 25000.0 if rate >= 2000000 else 1200.0
    iq = sig.make_batch("2fsk1k", 2, nframes=1, device_rate=rate, rx_offset_hz=offset, seed=43)
    n = iq.shape[1] & ~1
    chunk = 100002
    dem = q.Demod(qrl_ctx, 18, batch=2, max_chunk=chunk, device_samp_rate=rate, carrier_offset_hz=offset, time_domain_samp_rate=sr, time_domain_filter_width=fw)
    dem.enable_time_domain()
    d = torch.from_numpy(iq).cuda()
    parts = [[], []]
    used = 0
    for s0 in range(0, n, chunk):
        part = d[:, s0:min(s0 + chunk, n)]
        if part.shape[1] & 1:
            part = part[:, :-1]
        used += part.shape[1]
        dem.process(part.contiguous())
        cnt, sc = dem.scope_counts.cpu().numpy(), dem.scope.cpu().numpy()
        for b in range(2):
            parts[b].append(sc[b, :cnt[b]].copy())
    dem.close()
    D = 1000000 // sr if sr else 10
    taps = orc.low_pass(1, 1000000, fw, fw) if fw > 0 else orc.low_pass(1, 1000000, sr // 2 - sr // 8, sr // 4) if sr else orc.low_pass(1, 1000000, 50000, 25000)
    for b in range(2):
        fe = orc.frontend(iq[b, :used], rate, offset)
        want = orc.decim_auto(fe, taps, D).view(np.float32) + np.float32(0)
        got = np.concatenate(parts[b]).view(np.float32) + np.float32(0)
        assert got.size == want.size and got.size > 100, (got.size, want.size)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "stream %d" % b
    with pytest.raises(q.QrlError):
        q.Demod(qrl_ctx, 18, batch=1, max_chunk=1000, time_domain_samp_rate=600000)


def _nbfm_with_tone(n, seed, tone, fs=1000000.0, gap=None):
    """NBFM carrier whose audio is a voice-band tone plus a sub-audible CTCSS tone (deviation ~ 15 %)"""
    rng = np.random.default_rng(seed)
    t = np.arange(n) / fs
    a = 0.5 * np.sin(2 * np.pi * 1000.0 * t) + (0.15 * np.sin(2 * np.pi * tone * t) if tone else 0.0)
    ph = 2 * np.pi * 2500.0 * np.cumsum(a) / fs
    x = 0.05 * np.exp(1j * ph) + 0.0005 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    if gap:
        x[gap[0]:gap[1]] = 0.0005 * (rng.standard_normal(gap[1] - gap[0]) + 1j * rng.standard_normal(gap[1] - gap[0]))
    return x.astype(np.complex64)


@pytest.mark.parametrize("chunk", [1 << 21, 200000, 33334])
def test_nbfm_ctcss_squelch_bit_exact(qrl_ctx, chunk):
    """gr_demod_nbfm::set_ctcss(88.5) (src/gr/gr_demod_nbfm.cpp:59-60, 97-123): ctcss_squelch_ff(8000, 88.5, 0.01, 8000, 160, true) between
    audio resampler and audio filter, band-pass audio filter.  Three receivers: the right tone (audio opens after the first 1 s
    block), another tone of the table (91.5 Hz: stays shut), no tone (stays shut); audio and its COUNT equal the oracle bit for bit,
    in one call and cut into calls; switching the block out again restores the constructor's graph."""
    import torch
    import qradiolink_amd as q
    n = 2500000
    iq = np.stack([_nbfm_with_tone(n, 1, 88.5), _nbfm_with_tone(n, 2, 91.5), _nbfm_with_tone(n, 3, 0.0)])
    dem = q.Demod(qrl_ctx, q.MODEM_NBFM5000, batch=3, max_chunk=min(chunk, n))
    dem.set_ctcss(88.5)
    out = q.collect(dem, torch.from_numpy(iq).cuda(), min(chunk, n) & ~1)
    refs = [orc.demod_analog(iq[b], "nbfm", filter_width=5000, ctcss=88.5) for b in range(3)]
    for b in range(3):
        got, want = out["audio"][b] + np.float32(0), refs[b]["audio"] + np.float32(0)
        assert got.size == want.size, (b, got.size, want.size)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), b
    assert refs[0]["audio"].size > 8000 and refs[1]["audio"].size == 0 and refs[2]["audio"].size == 0
    a = refs[0]["audio"][2000:10000].astype(np.float64)
    spec = np.abs(np.fft.rfft(a * np.hanning(a.size)))
    assert abs(np.argmax(spec) * 8000.0 / a.size - 1000.0) < 3.0          # the voice tone; the 88.5 Hz tone is below the 300 Hz band-pass
    dem.set_ctcss(0.0)                                                      # back to the constructor's graph (fresh state)
    out2 = q.collect(dem, torch.from_numpy(iq).cuda(), min(chunk, n) & ~1)
    dem.close()
    for b in range(3):
        want = orc.demod_analog(iq[b], "nbfm", filter_width=5000)["audio"] + np.float32(0)
        got = out2["audio"][b] + np.float32(0)
        assert got.size == want.size and np.array_equal(got.view(np.uint32), want.view(np.uint32)), b


@pytest.mark.parametrize("chunk", [65536, 10007 * 2, 300002])
def test_chunk_invariance_gmsk(qrl_ctx, chunk):
    iq, out = _run(qrl_ctx, "gmsk10k", 22, 4000000, 25000.0, B=2, chunk=chunk, nframes=2)
    _compare(iq, out, "gmsk10k", 4000000, 25000.0)


@pytest.mark.parametrize("chunk", [1 << 20, 333334, 6400 * 3 + 2])
def test_chunk_invariance_mfma_front_end(qrl_ctx, chunk):
    """25 Msps front end (MFMA decimator): tiles are aligned to absolute output index, so any cut works."""
    iq, out = _run(qrl_ctx, "gmsk10k", 22, 25000000, 25000.0, B=2, chunk=chunk, nframes=1)
    _compare(iq, out, "gmsk10k", 25000000, 25000.0)


@pytest.mark.parametrize("chunk", [65536, 50000])
def test_chunk_invariance_2fsk(qrl_ctx, chunk):
    iq, out = _run(qrl_ctx, "2fsk1k", 18, 1000000, 1200.0, B=2, chunk=chunk, nframes=2)
    _compare(iq, out, "2fsk1k", 1000000, 1200.0)


@pytest.mark.parametrize("mode_name,modem,chunk", [("4fsk2k", 3, 44444), ("qpsk2k", 7, 60000), ("qpsk20k", 1, 33334), ("4fsk2kfm", 5, 50000), ("4fsk100k", 27, 20002), ("bpsk1k", 24, 65536),
                                                   ("bpsk2k", 0, 30000)])
def test_chunk_invariance_4fsk_bpsk(qrl_ctx, mode_name, modem, chunk):
    iq, out = _run(qrl_ctx, mode_name, modem, 1000000, 700.0, B=2, chunk=chunk, nframes=2)
    _compare(iq, out, mode_name, 1000000, 700.0)


@pytest.mark.parametrize("chunk", [65536, 20002, 1000])
def test_chunk_invariance_qpsk(qrl_ctx, chunk):
    iq, out = _run(qrl_ctx, "qpsk250k", 26, 1000000, 1200.0, B=2, chunk=chunk, nframes=2)
    _compare(iq, out, "qpsk250k", 1000000, 1200.0)


@pytest.mark.parametrize("mode_name,modem,rate", [
    ("2fsk1k", 18, 1000000),       # k_decim_pm: the edge outputs read the history rotated with the OLD offset, the buffer head with the new one
    ("gmsk10k", 22, 25000000),     # k_decim_pm, 25:1
    ("qpsk250k", 26, 100000000),   # k_decim_pm, 100:1
    ("gmsk10k", 22, 4000000),      # k_decim
])
def test_carrier_offset_retune_is_phase_continuous(qrl_ctx, mode_name, modem, rate):
    """qrl_demod_set_carrier_offset between calls = rotator_cc::set_phase_inc (gr_demod_base.cpp:1220-1225): the phase runs on, the
    increment changes from the next sample, and the samples already inside the decimator's history keep their old rotation.
    Expected: the oracle chain (offset 0) on the input rotated piecewise by the oracle's exact NCO."""
    import torch
    import qradiolink_amd as q
    f1, f2 = (1200.0, -800.0) if rate < 2000000 else (25000.0, -12500.0)
    iq = sig.make_batch(mode_name, 2, nframes=2, device_rate=rate, rx_offset_hz=f1, seed=13)
    n1 = (iq.shape[1] // 3) & ~1
    dem = q.Demod(qrl_ctx, modem, batch=2, max_chunk=iq.shape[1], device_samp_rate=rate, carrier_offset_hz=f1)
    d = torch.from_numpy(iq).cuda()
    parts = []
    for lo, hi, f in ((0, n1, None), (n1, iq.shape[1], f2)):
        if f is not None:
            q._check(dem.lib.qrl_demod_set_carrier_offset(dem.h, f), "qrl_demod_set_carrier_offset")
        o = dem.process(d[:, lo:hi].contiguous())
        c = o["counts"].cpu().numpy()
        parts.append({k: [o[k][b, :c[b, j]].cpu().numpy().copy() for b in range(2)] for k, j in (("filtered", 0), ("bits_a", 2))})
    dem.close()
    inc1 = orc.phase_inc_to_turn(2 * np.pi * -f1 / rate)
    inc2 = orc.phase_inc_to_turn(2 * np.pi * -f2 / rate)
    for b in range(2):
        y = np.concatenate([orc.rotator(iq[b, :n1], inc1, 0), orc.rotator(iq[b, n1:], inc2, (n1 * inc1) & 0xFFFFFFFFFFFFFFFF)])
        # the oracle front end with offset 0 multiplies by the exact phasor (1, 0): an identity
        ref = _oracle(mode_name, y, rate, 0.0)
        got_f = np.concatenate([p["filtered"][b] for p in parts]).view(np.float32) + np.float32(0)
        got_a = np.concatenate([p["bits_a"][b] for p in parts])
        assert np.array_equal(got_f.view(np.uint32), (ref["filtered"].view(np.float32) + np.float32(0)).view(np.uint32))
        assert got_a.size == ref["bits_a"].size and np.array_equal(got_a, ref["bits_a"])


def test_front_end_repeatable_under_load(qrl_ctx):
    """Race hunt: the 25 Msps front end runs two workgroups per CU; repeat the same batch and require
    bit-identical port 0 every time (an earlier hand-scheduled LDS pipeline failed this ~1e-4 per tile)."""
    import torch
    import qradiolink_amd as q
    rate, offset, B = 25000000, 25000.0, 6
    iq = sig.make_batch("gmsk10k", B, nframes=2, device_rate=rate, rx_offset_hz=offset, seed=3)
    d = torch.from_numpy(iq).cuda()
    ref = None
    for rep in range(6):
        dem = q.Demod(qrl_ctx, q.MODEM_GMSK10K, batch=B, max_chunk=1 << 23, device_samp_rate=rate, carrier_offset_hz=offset)
        out = q.collect(dem, d, 1 << 23)
        dem.close()
        got = [x.view(np.uint32).copy() for x in out["filtered"]]
        if ref is None:
            ref = got
            want = orc.demod_gmsk(orc.frontend(iq[0], rate, offset), sps=1, filter_width=20000)["filtered"].view(np.uint32)
            assert np.array_equal(got[0], want)
        else:
            for b in range(B):
                assert np.array_equal(got[b], ref[b]), "run %d stream %d differs from run 0" % (rep, b)


def test_front_end_repeatable_200_runs_at_bench_like_batch(qrl_ctx):
    """The same race hunt where it matters: 512 streams (two workgroups per CU on every CU, tiles-per-workgroup > 1, the asm-issued
    register prefetch of tile_issue / tile_wait in flight everywhere), 200 repetitions of the same call from a fresh state; every
    repetition must reproduce run 0 bit for bit on all four ports, and run 0 equals the oracle on a few streams.  (The ISA-level
    proof that no instruction touches a prefetch register before its wait is tests/test_isa_audit.py.)"""
    import torch
    import qradiolink_amd as q
    rate, offset, B = 25000000, 25000.0, 512
    base = sig.make_batch("gmsk10k", 4, nframes=1, device_rate=rate, rx_offset_hz=offset, seed=13)
    n = base.shape[1] & ~1
    idx = np.arange(B) % 4
    d = torch.from_numpy(base[:, :n]).cuda()[torch.from_numpy(idx).cuda()].contiguous()
    dem = q.Demod(qrl_ctx, q.MODEM_GMSK10K, batch=B, max_chunk=n, device_samp_rate=rate, carrier_offset_hz=offset)
    ref = None
    for rep in range(200):
        dem.reset()
        out = dem.process(d)
        cur = [out[k].clone() for k in ("filtered", "constellation", "bits_a", "bits_b", "counts")]
        if ref is None:
            ref = cur
            cnt = cur[4].cpu().numpy()
            for b in (0, 1, 2, 3, 257, 511):
                want = _oracle("gmsk10k", base[idx[b], :n], rate, offset)
                assert np.array_equal(cur[2][b, :cnt[b, 2]].cpu().numpy(), want["bits_a"])
                got = cur[0][b, :cnt[b, 0]].cpu().numpy().view(np.float32) + np.float32(0)
                assert np.array_equal(got.view(np.uint32), (want["filtered"].view(np.float32) + np.float32(0)).view(np.uint32))
        else:
            for k in range(5):
                assert torch.equal(torch.view_as_real(cur[k]) if cur[k].is_complex() else cur[k],
                                   torch.view_as_real(ref[k]) if ref[k].is_complex() else ref[k]), "run %d differs from run 0 (port %d)" % (rep, k)
    dem.close()


def test_loopback_frames_recovered(qrl_ctx):
    """mod -> channel -> HIP demod returns the transmitted frames through gr_modem-style sync search."""
    import torch
    import qradiolink_amd as q
    y, payloads = sig.make_stream("gmsk10k", nframes=4, device_rate=1000000, seed=5)
    y = y[: y.size & ~1]
    dem = q.Demod(qrl_ctx, q.MODEM_GMSK10K, batch=1, max_chunk=y.size)
    out = q.collect(dem, torch.from_numpy(y[None, :]).cuda(), y.size)
    dem.close()
    got = [sum((bytes([0xAA]) + p) in sig.find_frames(out[k][0], bytes([0xED, 0x89]), 384) for p in payloads) for k in ("bits_a", "bits_b")]
    assert max(got) == len(payloads), got


# ---- DMR / 4FSK symbol demodulator (gr_demod_dmr, the in-tree "4FSK demod" of BASELINE config 4)
@pytest.mark.parametrize("chunk", [1 << 20, 50000, 12502])
def test_dmr_4fsk_bit_exact(qrl_ctx, chunk):
    import torch
    import qradiolink_amd as q
    xs = [sig.make_4fsk(nsym=300, seed=s)[0] for s in (1, 2)]
    n = min(x.size for x in xs)
    iq = np.stack([x[:n] for x in xs])
    dem = q.Demod(qrl_ctx, q.MODEM_DMR, batch=2, max_chunk=chunk)
    out = q.collect(dem, torch.from_numpy(iq).cuda(), chunk)
    dem.close()
    for b in range(2):
        ref = orc.demod_dmr(iq[b])
        assert np.array_equal(out["bits_a"][b], ref["bits_a"]) and ref["bits_a"].size > 500
        for port in ("filtered", "constellation"):
            assert np.array_equal(out[port][b].view(np.uint32), ref[port].view(np.uint32)), port


def test_dmr_dibits_recovered(qrl_ctx):
    import torch
    import qradiolink_amd as q
    x, dib = sig.make_4fsk(nsym=500, seed=7)
    dem = q.Demod(qrl_ctx, q.MODEM_DMR, batch=1, max_chunk=x.size)
    out = q.collect(dem, torch.from_numpy(x[None, :]).cuda(), x.size)
    dem.close()
    got = out["bits_a"][0].reshape(-1, 2)
    got = got[:, 0] * 2 + got[:, 1]
    best = max(np.mean(got[k:k + 400] == dib[:400]) for k in range(40))
    assert best == 1.0


# ---- M17 4FSK symbol demodulator (gr_demod_m17, SURVEY 8(f) rank 4: same kernels as gr_demod_dmr + a channel filter, mod-M&M TED)
@pytest.mark.parametrize("chunk", [1 << 20, 50000, 12502])
def test_m17_4fsk_bit_exact(qrl_ctx, chunk):
    import torch
    import qradiolink_amd as q
    xs = [sig.make_4fsk(nsym=300, seed=s, alpha=0.5, dev=2400.0)[0] for s in (4, 5)]
    n = min(x.size for x in xs)
    iq = np.stack([x[:n] for x in xs])
    dem = q.Demod(qrl_ctx, q.MODEM_M17, batch=2, max_chunk=chunk)
    out = q.collect(dem, torch.from_numpy(iq).cuda(), chunk)
    dem.close()
    for b in range(2):
        ref = orc.demod_m17(iq[b])
        assert np.array_equal(out["bits_a"][b], ref["bits_a"]) and ref["bits_a"].size > 500
        for port in ("filtered", "constellation"):
            assert np.array_equal(out[port][b].view(np.uint32), ref[port].view(np.uint32)), port


def test_m17_dibits_recovered(qrl_ctx):
    import torch
    import qradiolink_amd as q
    x, dib = sig.make_4fsk(nsym=500, seed=9, alpha=0.5, dev=2400.0)
    dem = q.Demod(qrl_ctx, q.MODEM_M17, batch=1, max_chunk=x.size)
    out = q.collect(dem, torch.from_numpy(x[None, :]).cuda(), x.size)
    dem.close()
    got = out["bits_a"][0].reshape(-1, 2)
    got = got[:, 0] * 2 + got[:, 1]
    # (the clock loop -- bandwidth 2 pi / 96, 25 x DMR's -- needs ~40 symbols to settle on a burst without preamble)
    best = max(np.mean(got[k + 60:k + 420] == dib[60:420]) for k in range(60))
    assert best == 1.0


# ---- analogue voice receivers (gr_demod_nbfm / gr_demod_am / gr_demod_wbfm): gating squelch, so the audio
# count is data dependent; a stretch of exact zeros (idle channel) closes the squelch in the middle of the stream
@pytest.mark.parametrize("kind,modem,fw", [("nbfm", 9, 5000), ("nbfm", 8, 2500), ("am", 14, 5000), ("wbfm", 10, 75000)])
@pytest.mark.parametrize("chunk", [1 << 20, 100000, 33334])
def test_analog_bit_exact(qrl_ctx, kind, modem, fw, chunk):
    import torch
    import qradiolink_amd as q
    n = 400000
    xs = [sig.make_analog(kind, n=n, seed=1, gap=(100000, 300000))[0], sig.make_analog(kind, n=n, seed=2)[0]]
    iq = np.stack(xs)
    dem = q.Demod(qrl_ctx, modem, batch=2, max_chunk=min(chunk, n))
    out = q.collect(dem, torch.from_numpy(iq).cuda(), min(chunk, n))
    dem.close()
    for b in range(2):
        ref = orc.demod_analog(iq[b], kind, filter_width=fw)
        assert ref["audio"].size > 1500
        got, want = out["filtered"][b].view(np.float32) + np.float32(0), ref["filtered"].view(np.float32) + np.float32(0)
        assert got.size == want.size and np.array_equal(got.view(np.uint32), want.view(np.uint32)), "filtered"
        got, want = out["audio"][b] + np.float32(0), ref["audio"] + np.float32(0)
        assert got.size == want.size, (got.size, want.size)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "audio"
    # the idle stretch of stream 0 is gated away: fewer audio samples than the stream without it
    assert out["audio"][0].size < out["audio"][1].size


@pytest.mark.parametrize("kind,modem", [("nbfm", 9), ("am", 14), ("wbfm", 10)])
def test_analog_audio_is_the_modulating_tone(qrl_ctx, kind, modem):
    import torch
    import qradiolink_amd as q
    x, _ = sig.make_analog(kind, n=600000, seed=1)
    dem = q.Demod(qrl_ctx, modem, batch=1, max_chunk=x.size)
    out = q.collect(dem, torch.from_numpy(x[None, :]).cuda(), x.size)
    dem.close()
    a = out["audio"][0][800:4000].astype(np.float64)
    spec = np.abs(np.fft.rfft(a * np.hanning(a.size)))
    peak_hz = (np.argmax(spec[5:]) + 5) * 8000.0 / a.size
    assert abs(peak_hz - 713.0) < 5.0, peak_hz


def test_analog_squelch_threshold(qrl_ctx):
    """set_squelch(db): a threshold above the signal's power keeps the gate shut -> no audio at all"""
    import torch
    import qradiolink_amd as q
    x, _ = sig.make_analog("nbfm", n=200000, seed=3)          # power 0.05^2 = -26 dB
    dem = q.Demod(qrl_ctx, q.MODEM_NBFM5000, batch=1, max_chunk=x.size)
    dem.set_squelch(-10.0)
    out = q.collect(dem, torch.from_numpy(x[None, :]).cuda(), x.size)
    assert out["audio"][0].size == 0 and out["filtered"][0].size == 4000
    dem.reset()
    dem.set_squelch(-140.0)
    out = q.collect(dem, torch.from_numpy(x[None, :]).cuda(), x.size)
    dem.close()
    # (the first items of the stream, where the channel filter is still filling, stay below even -140 dB)
    assert out["audio"][0].size == orc.demod_analog(x, "nbfm", filter_width=5000)["audio"].size and 1590 <= out["audio"][0].size <= 1600


@pytest.mark.parametrize("lsb,modem", [(False, 11), (True, 12)])
@pytest.mark.parametrize("chunk", [1 << 20, 150000, 33334])
def test_ssb_bit_exact(qrl_ctx, lsb, modem, chunk):
    """gr_demod_ssb: 1:125, IF gain, complex band-pass, gating squelch, agc2_cc, cessb clipper + stretcher (whole chunks of 1024
    gated items, two items of look-ahead), real part, audio band-pass"""
    import torch
    import qradiolink_amd as q
    n = 1200000
    xs = [sig.make_ssb(n=n, seed=1, lsb=lsb, gap=(300000, 900000)), sig.make_ssb(n=n, seed=2, lsb=lsb)]
    iq = np.stack(xs)
    dem = q.Demod(qrl_ctx, modem, batch=2, max_chunk=min(chunk, n))
    out = q.collect(dem, torch.from_numpy(iq).cuda(), min(chunk, n))
    dem.close()
    for b in range(2):
        ref = orc.demod_ssb(iq[b], sb=int(lsb))
        assert ref["audio"].size >= 4096 and ref["audio"].size % 1024 == 0
        got, want = out["filtered"][b].view(np.float32) + np.float32(0), ref["filtered"].view(np.float32) + np.float32(0)
        assert got.size == want.size and np.array_equal(got.view(np.uint32), want.view(np.uint32)), "filtered"
        got, want = out["audio"][b] + np.float32(0), ref["audio"] + np.float32(0)
        assert got.size == want.size, (got.size, want.size)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "audio"
    assert out["audio"][0].size < out["audio"][1].size


def test_ssb_audio_is_the_modulating_tone_and_the_other_sideband_is_rejected(qrl_ctx):
    import torch
    import qradiolink_amd as q
    x = sig.make_ssb(n=800000, seed=1, lsb=False)
    rms = {}
    for modem in (q.MODEM_USB2500, q.MODEM_LSB2500):
        dem = q.Demod(qrl_ctx, modem, batch=1, max_chunk=x.size)
        out = q.collect(dem, torch.from_numpy(x[None, :]).cuda(), x.size)
        dem.close()
        a = out["audio"][0][1024:5120].astype(np.float64)
        rms[modem] = np.sqrt(np.mean(a ** 2))
        if modem == q.MODEM_USB2500:
            spec = np.abs(np.fft.rfft(a * np.hanning(a.size)))
            assert abs(np.argmax(spec) * 8000.0 / a.size - 713.0) < 4.0
    assert rms[q.MODEM_LSB2500] < 0.02 * rms[q.MODEM_USB2500]


# ---- DSSS "BPSK 8" (gr_demod_dsss): 1:50, 13:50 resampler, Costas, filter, AGC, Barker-13 matched-filter
# decoder (325 evaluations of a 600-tap filter per symbol), M&M clock recovery, Costas, K=7 decoder on two branches
@pytest.mark.parametrize("chunk", [1 << 22, 250000, 65538])
def test_dsss_bit_exact(qrl_ctx, chunk):
    import torch
    import qradiolink_amd as q
    rng = np.random.default_rng(5)
    xs = [sig.make_dsss(rng.integers(0, 2, 60, dtype=np.uint8), seed=s, cfo=c) for s, c in ((1, 0.0), (2, 3.0))]
    n = min(x.size for x in xs) & ~1
    iq = np.stack([x[:n] for x in xs])
    dem = q.Demod(qrl_ctx, q.MODEM_BPSK8, batch=2, max_chunk=min(chunk, n))
    out = q.collect(dem, torch.from_numpy(iq).cuda(), min(chunk, n))
    dem.close()
    for b in range(2):
        ref = orc.demod_dsss(iq[b])
        for port in ("bits_a", "bits_b"):
            assert np.array_equal(out[port][b], ref[port]), port
        assert ref["constellation"].size > 100
        for port in ("filtered", "constellation"):
            got, want = out[port][b].view(np.float32) + np.float32(0), ref[port].view(np.float32) + np.float32(0)
            assert got.size == want.size, (port, got.size, want.size)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), port


def test_dsss_info_bits_recovered(qrl_ctx):
    import torch
    import qradiolink_amd as q
    bits = np.random.default_rng(3).integers(0, 2, 120, dtype=np.uint8)
    x = sig.make_dsss(bits)
    x = x[:x.size & ~1]
    dem = q.Demod(qrl_ctx, q.MODEM_BPSK8, batch=1, max_chunk=x.size)
    out = q.collect(dem, torch.from_numpy(x[None, :]).cuda(), x.size)
    dem.close()
    want = "".join(map(str, bits[:60]))
    assert any(want in "".join(map(str, out[p][0])) for p in ("bits_a", "bits_b"))


# the Viterbi decoder away from a clean signal (round 6: k_fec rebuilt with a rotating state -> lane layout): noise only -- every path metric close
# to every other, ties at every step -- and inputs so strong or so weak that the soft symbols sit at 0 / 255 or at 128 (saturated adds,
# renormalisation at almost every step, or never); cc_decoder's block chaining (the start state is the state six steps before the end of
# the block before) across call cuts that do not line up with the 80-bit blocks
@pytest.mark.parametrize("mode_name,modem,rate", [("2fsk1k", 18, 1000000), ("gmsk10k", 22, 1000000), ("qpsk250k", 26, 1000000), ("bpsk2k", 0, 1000000),
                                                   ("4fsk100k", 27, 1000000)])
@pytest.mark.parametrize("amp", [1e-4, 0.05, 40.0])
def test_decoder_on_noise_and_saturating_inputs(qrl_ctx, mode_name, modem, rate, amp):
    import torch
    import qradiolink_amd as q
    rng = np.random.default_rng(int(amp * 1e4) + modem)
    n = 300000 if mode_name in ("qpsk250k", "4fsk100k", "gmsk10k") else 1200000
    noise = (rng.standard_normal((2, n)) + 1j * rng.standard_normal((2, n))).astype(np.complex64) * np.float32(amp)
    sigl = sig.make_batch(mode_name, 1, nframes=1, device_rate=rate, rx_offset_hz=1200.0, seed=9)[0]
    iq = noise.copy()
    m = min(n, sigl.size)
    iq[1, :m] += sigl[:m] * np.float32(amp / 0.05)          # stream 1: the frame under the same noise level (about 0 dB)
    dem = q.Demod(qrl_ctx, modem, batch=2, max_chunk=77777, device_samp_rate=rate, carrier_offset_hz=1200.0)
    out = q.collect(dem, torch.from_numpy(iq).cuda(), 77777)
    dem.close()
    _compare(iq, out, mode_name, rate, 1200.0)


@pytest.mark.parametrize("fw", [3000, 1500])
@pytest.mark.parametrize("chunk", [1 << 21, 70000])
def test_2fsk_explicit_filter_width_runs_the_run_time_tap_loop(qrl_ctx, fw, chunk):
    """gr_demod_2fsk with a filter width other than the modes' own (explicit configuration: qrl_demod_config.use_mode_defaults = 0): other tap
    counts than 41 + 41 + 25, so k_2fsk_ff takes its run-time tap loop instead of the <11, 11, 7> instantiation (3000 Hz: shorter filters;
    1500 Hz: 53-tap filters, the unfused kernels) -- bit-exact against the oracle chain with the same width."""
    import torch
    import qradiolink_amd as q
    iq = sig.make_batch("2fsk1k", 3, nframes=2, device_rate=1000000, rx_offset_hz=1200.0, seed=17, impair=sig.SPEC)
    dem = q.Demod(qrl_ctx, 18, batch=3, max_chunk=chunk, device_samp_rate=1000000, carrier_offset_hz=1200.0, sps=10, filter_width=fw)
    out = q.collect(dem, torch.from_numpy(iq).cuda(), chunk)
    dem.close()
    for b in range(3):
        ref = orc.demod_2fsk(orc.frontend(iq[b], 1000000, 1200.0), sps=10, filter_width=fw, fm=False)
        assert ref["bits_a"].size > 0
        for port in ("bits_a", "bits_b"):
            assert np.array_equal(out[port][b], ref[port]), (port, b)
        for port in ("filtered", "constellation"):
            got, want = out[port][b].view(np.float32) + np.float32(0), ref[port].view(np.float32) + np.float32(0)
            assert got.size == want.size and np.array_equal(got.view(np.uint32), want.view(np.uint32)), (port, b)


@pytest.mark.parametrize("mode_name,modem,rate", [("2fsk1k", 18, 1000000), ("gmsk10k", 22, 4000000), ("qpsk250k", 26, 10000000), ("gmsk10k", 22, 25000000)])
def test_front_end_helper_stream_and_input_buffer_reuse(qrl_ctx, mode_name, modem, rate):
    """Round 6: k_hist (the tail of a call's IQ kept for the next call) and k_pl_edge_stage (the next call's edge scratch) run on a helper
    stream beside the front end when the handle owns its streams; a handle on a CALLER's stream keeps them in line.  Both orders give the
    oracle's bits -- with every call's input in ONE device buffer that is overwritten as soon as qrl_demod_stream_wait lets the copy stream
    go on (the helper kernels read the caller's buffer: the wait has to cover them), and no qrl_demod_sync between the calls."""
    import torch
    import qradiolink_amd as q
    B = 6
    offset = 25000.0 if rate >= 2000000 else 1200.0
    iq = sig.make_batch(mode_name, B, nframes=2, device_rate=rate, rx_offset_hz=offset, seed=12)
    chunk = 50000 * (rate // 1000000)
    ncalls = iq.shape[1] // chunk
    host = torch.from_numpy(iq[:, :ncalls * chunk].copy()).pin_memory()
    refs = [_oracle(mode_name, iq[b, :ncalls * chunk], rate, offset)["bits_a"] for b in range(B)]
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    for order in ("resident", "inline", "caller"):
        # resident: QRL_OPT_INPUT_RESIDENT = 1, the copy is complete before the call.  inline: the option off and the upload QUEUED ON THE
        # HANDLE'S STREAM right before the call, not waited for (qrl_host::gr_demod_base_hip::work does that) -- everything that reads the
        # buffer has to sit behind the handle's stream.  caller: the handle on a stream of the caller's.
        user, copy = torch.cuda.Stream(), torch.cuda.Stream()
        dem = q.Demod(qrl_ctx, modem, batch=B, max_chunk=chunk, device_samp_rate=rate, carrier_offset_hz=offset,
                      stream=user.cuda_stream if order == "caller" else None, input_resident=order == "resident")
        buf = torch.zeros((B, chunk), dtype=torch.complex64, device="cuda")
        nbytes = B * chunk * 8
        for k in range(ncalls):
            src = host[:, k * chunk:(k + 1) * chunk].contiguous().pin_memory()
            if order == "inline":
                # behind the reads of call k - 1 (the handle's stream waits for the copy stream, which waited for the handle), then the upload on the handle's stream
                ev = torch.cuda.Event(); ev.record(copy)
                assert hip.hipStreamWaitEvent(ctypes.c_void_p(dem.stream), ctypes.c_void_p(ev.cuda_event), 0) == 0
                assert hip.hipMemcpyAsync(ctypes.c_void_p(buf.data_ptr()), ctypes.c_void_p(src.data_ptr()), ctypes.c_size_t(nbytes), 1, ctypes.c_void_p(dem.stream)) == 0
            else:
                with torch.cuda.stream(copy):
                    buf.copy_(src, non_blocking=True)          # overwrites what call k - 1 read: behind its stream_wait
                copy.synchronize()                             # (the host waits for the COPY only; the copy waited on the device for the handle's reads)
            dem.process_async(buf)
            dem.stream_wait(copy.cuda_stream)
            if order == "inline":
                torch.cuda.synchronize()                       # `src` (a fresh pinned staging tensor per call) must outlive the upload
        dem.sync()
        cnt, a = dem.counts.cpu().numpy(), dem.bits_a.cpu().numpy()
        dem.close()
        for b in range(B):
            n_last = int(cnt[b, 2])
            assert refs[b].size > 0 and n_last > 0
            assert np.array_equal(a[b, :n_last], refs[b][-n_last:]), "order %s, stream %d" % (order, b)


@pytest.mark.parametrize("mode_name,modem", [("2fsk1k", 18), ("2fsk1kfm", 16), ("gmsk10k", 22), ("qpsk250k", 26), ("bpsk2k", 0)])
def test_pipelined_calls_without_sync(qrl_ctx, mode_name, modem):
    """Back-to-back qrl_demod_process calls with NO sync in between (how bench.py drives the handle; the 2FSK family then runs
    its decimated-rate kernels of call k under the front end of call k + 1, ring s2 holding two calls; the QPSK / BPSK families run
    feed-forward kernels, recursion and Viterbi decoder of three consecutive calls side by side on three streams): the outputs of
    the last call equal those of the same sequence run with a sync after every call, and the bits equal the oracle's tail."""
    import torch
    import qradiolink_amd as q
    B, chunk = 130, 40000
    iq = sig.make_batch(mode_name, 2, nframes=2, device_rate=1000000, seed=5)
    iq = np.concatenate([iq, iq[::-1]] * (B // 4 + 1))[:B]
    ncalls = iq.shape[1] // chunk
    d = torch.from_numpy(iq).cuda()
    res = []
    for sync_each in (True, False):
        dem = q.Demod(qrl_ctx, modem, batch=B, max_chunk=chunk)
        if mode_name.startswith("2fsk"):
            # QRL_OPT_OVERLAP is the library default for this family since round 3: the synced reference run takes the SERIAL order
            # (cs = stream, ev_tail / tail_pending waits), the unsynced run the overlapped one -- both paths are checked against each
            # other and against the oracle
            dem.set_option(q.OPT_OVERLAP, 0 if sync_each else 1)
        if mode_name == "qpsk250k":
            # QRL_OPT_GROUPED (round 4; the default at chip-filling batches): the decoder of call k is launched with the recursion of
            # call k + 1 (or by the next function that waits for results) -- forced on in the unsynced run, off in the synced one
            dem.set_option(q.OPT_GROUPED, 0 if sync_each else 1)
        for k in range(ncalls):
            dem.process_async(d[:, k * chunk:(k + 1) * chunk])
            if sync_each:
                dem.sync()
        dem.sync()
        res.append((dem.counts.cpu().numpy().copy(), dem.bits_a.cpu().numpy().copy(), dem.bits_b.cpu().numpy().copy()))
        dem.close()
    (c0, a0, b0), (c1, a1, b1) = res
    assert np.array_equal(c0, c1)
    for b in range(B):
        assert np.array_equal(a0[b, :c0[b, 2]], a1[b, :c1[b, 2]]) and np.array_equal(b0[b, :c0[b, 3]], b1[b, :c1[b, 3]])
    ref = _oracle(mode_name, iq[0, :ncalls * chunk], 1000000, 0.0)
    n_last = int(c1[0, 2])
    assert n_last > 0 and np.array_equal(a1[0, :n_last], ref["bits_a"][-n_last:])


def test_grouped_order_at_a_chip_filling_batch(qrl_ctx):
    """The QPSK receiver at a batch that gives k_qpsk_pipe4 a workgroup for every second CU (QRL_OPT_GROUPED's default there): calls queued
    back to back run front end -> recursion || decoder of the call before (the decoder launched one call late, behind k_fec_gate); the
    same sequence with a sync after every call flushes every decoder at once.  Every stream's counts and bits must agree, and two
    streams equal the oracle."""
    import torch
    import qradiolink_amd as q
    B, chunk, ncalls = 8200, 6000, 3
    base = sig.make_batch("qpsk250k", 2, nframes=2, device_rate=1000000, seed=12)[:, :chunk * ncalls]
    d = torch.from_numpy(base).cuda()
    d = torch.cat([d, torch.flip(d, [0])]).repeat(B // 4 + 1, 1)[:B].contiguous()
    res = []
    for sync_each in (True, False):
        dem = q.Demod(qrl_ctx, 26, batch=B, max_chunk=chunk)
        outs = []
        for k in range(ncalls):
            outs.append(dem.new_outputs())
            dem.process_async(d[:, k * chunk:(k + 1) * chunk])
            if sync_each:
                dem.sync()
        dem.sync()
        res.append([(o["counts"].cpu().numpy().copy(), o["bits_a"].cpu().numpy().copy()) for o in outs])
        dem.close()
    total = 0
    for k in range(ncalls):
        (c0, a0), (c1, a1) = res[0][k], res[1][k]
        assert np.array_equal(c0, c1), "call %d" % k
        n = int(c0[:, 2].max())
        mask = np.arange(n)[None, :] < c0[:, 2:3]
        assert np.array_equal(a0[:, :n][mask], a1[:, :n][mask]), "call %d" % k
        total += int(c0[:, 2].sum())
    assert total > 0
    for b in (0, 1):
        ref = _oracle("qpsk250k", base[b], 1000000, 0.0)
        got = np.concatenate([res[1][k][1][b, :res[1][k][0][b, 2]] for k in range(ncalls)])
        assert np.array_equal(got, ref["bits_a"][:got.size]) and got.size > 0


@pytest.mark.parametrize("mode_name,modem", [("qpsk250k", 26), ("bpsk2k", 0), ("gmsk10k", 22), ("2fsk1k", 18)])
def test_every_call_of_a_pipelined_sequence_is_deterministic(qrl_ctx, mode_name, modem):
    """Calls in flight together (each with its own output buffers, no sync in between): EVERY call's counts, bits and port 1 equal
    those of the same sequence run with a sync after every call -- the Viterbi decoder of call k, running beside the recursion of
    call k + 1, must only see call k's symbols (per-call symbol-count snapshot), whatever the timing."""
    import torch
    import qradiolink_amd as q
    B, chunk = 70, 30000
    iq = sig.make_batch(mode_name, 2, nframes=2, device_rate=1000000, seed=8)
    iq = np.concatenate([iq, iq[::-1]] * (B // 4 + 1))[:B]
    ncalls = iq.shape[1] // chunk
    d = torch.from_numpy(iq).cuda()
    runs = []
    for sync_each in (True, False):
        dem = q.Demod(qrl_ctx, modem, batch=B, max_chunk=chunk)
        if mode_name.startswith("2fsk"):
            dem.set_option(q.OPT_OVERLAP, 0 if sync_each else 1)   # serial order in the synced run, overlapped (the default) in the other
        if mode_name == "qpsk250k":
            dem.set_option(q.OPT_GROUPED, 0 if sync_each else 1)   # deferred decoder launches in the unsynced run: every call still gets ITS bits
        outs = []
        for k in range(ncalls):
            outs.append(dem.new_outputs())
            dem.process_async(d[:, k * chunk:(k + 1) * chunk])
            if sync_each:
                dem.sync()
        dem.sync()
        runs.append([{key: v.cpu().numpy().copy() for key, v in o.items() if v is not None} for o in outs])
        dem.close()
    total_bits = 0
    for k in range(ncalls):
        r0, r1 = runs[0][k], runs[1][k]
        assert np.array_equal(r0["counts"], r1["counts"]), "call %d" % k
        for b in range(B):
            c = r0["counts"][b]
            assert np.array_equal(r0["bits_a"][b, :c[2]], r1["bits_a"][b, :c[2]]) and np.array_equal(r0["bits_b"][b, :c[3]], r1["bits_b"][b, :c[3]])
            assert np.array_equal(r0["constellation"][b, :c[1]].view(np.uint32), r1["constellation"][b, :c[1]].view(np.uint32))
        total_bits += int(r0["counts"][:, 2].sum())
    assert total_bits > 0


@pytest.mark.parametrize("mode_name,modem,rate,B,nframes", [
    ("2fsk1k", 18, 1000000, 1536, 1),       # C1 at a bench-like batch: several segments per stream, grid.x >> 8 (k_decim_pm units, FLL quads)
    ("gmsk10k", 22, 25000000, 1024, 1),     # C2 at a bench-like batch: MFMA front end with tpw > 1 and the XCD remap of grid.x
])
def test_large_batch_bit_exact(qrl_ctx, mode_name, modem, rate, B, nframes):
    """Grids, tiles-per-workgroup and stream indexing at bench-like batch sizes: B streams cycle through 6 distinct seeded
    inputs; every one of the B outputs must equal the oracle's output for its input."""
    import torch
    import qradiolink_amd as q
    offset = 25000.0 if rate >= 2000000 else 1200.0
    base = sig.make_batch(mode_name, 6, nframes=nframes, device_rate=rate, rx_offset_hz=offset, seed=21)
    idx = np.arange(B) % 6
    iq = torch.from_numpy(base).cuda()[torch.from_numpy(idx).cuda()].contiguous()
    dem = q.Demod(qrl_ctx, modem, batch=B, max_chunk=base.shape[1], device_samp_rate=rate, carrier_offset_hz=offset)
    out = q.collect(dem, iq, base.shape[1])
    dem.close()
    refs = [_oracle(mode_name, base[k], rate, offset) for k in range(6)]
    for b in range(B):
        ref = refs[idx[b]]
        assert np.array_equal(out["bits_a"][b], ref["bits_a"]) and np.array_equal(out["bits_b"][b], ref["bits_b"]), "bits of stream %d" % b
        got, want = out["filtered"][b].view(np.float32) + np.float32(0), ref["filtered"].view(np.float32) + np.float32(0)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "port 0 of stream %d" % b


@pytest.mark.parametrize("pitch_pad", [0, 6, 130])
def test_lds_dma_front_end_with_padded_row_pitch(qrl_ctx, pitch_pad):
    """k_decim_pm cuts every stream row into 16-byte LDS-DMA pieces relative to the row start and clamps the last piece at the row
    end: rows whose pitch is not a multiple of 128 bytes, and the last row of the allocation, must come out the same."""
    import torch
    import qradiolink_amd as q
    iq = sig.make_batch("2fsk1k", 4, nframes=2, device_rate=1000000, rx_offset_hz=1200.0, seed=5)
    n = iq.shape[1]
    buf = torch.zeros((4, n + pitch_pad), dtype=torch.complex64, device="cuda")
    buf[:, :n] = torch.from_numpy(iq).cuda()
    view = buf[:, :n]
    dem = q.Demod(qrl_ctx, 18, batch=4, max_chunk=n, device_samp_rate=1000000, carrier_offset_hz=1200.0)
    out = q.collect(dem, view, n)
    dem.close()
    _compare(iq, out, "2fsk1k", 1000000, 1200.0)


@pytest.mark.parametrize("chunk", [1 << 20, 4050, 7778])
def test_qpsk250k_unfused_resampler_and_filter_agree_with_the_fused_kernel(qrl_ctx, chunk):
    """QRL_OPT_UNFUSED_DEC2 = 1 runs the 1:2 resampler and the RRC of gr_demod_qpsk (gr_demod_qpsk.cpp:92-103) as the two kernels of
    rounds 1-2; the fused k_dec2_fir (default) must give the same bits AND the same port-0 / port-1 floats, both equal to the oracle's.
    Chunk sizes around the kernel's 2024-output tile (4048 input samples) exercise the halo taken from the carried history."""
    import torch
    import qradiolink_amd as q
    iq = sig.make_batch("qpsk250k", 2, nframes=2, device_rate=1000000, rx_offset_hz=1200.0, seed=11)
    outs = []
    for unfused in (0, 1):
        dem = q.Demod(qrl_ctx, 26, batch=2, max_chunk=chunk, device_samp_rate=1000000, carrier_offset_hz=1200.0)
        dem.set_option(q.OPT_UNFUSED_DEC2, unfused)
        outs.append(q.collect(dem, torch.from_numpy(iq).cuda(), chunk))
        dem.close()
    _compare(iq, outs[0], "qpsk250k", 1000000, 1200.0)
    _compare(iq, outs[1], "qpsk250k", 1000000, 1200.0)


def test_unfused_option_is_refused_mid_stream_and_on_other_chains(qrl_ctx):
    import torch
    import qradiolink_amd as q
    dem = q.Demod(qrl_ctx, 18, batch=1, max_chunk=4096, device_samp_rate=1000000)
    with pytest.raises(q.QrlError):
        dem.set_option(q.OPT_UNFUSED_DEC2, 1)
    dem.close()
    dem = q.Demod(qrl_ctx, 26, batch=1, max_chunk=4096, device_samp_rate=1000000)
    dem.process(torch.zeros((1, 4096), dtype=torch.complex64, device="cuda"))
    with pytest.raises(q.QrlError):
        dem.set_option(q.OPT_UNFUSED_DEC2, 1)
    dem.close()


# ---- gr_demod_base::set_filter_width / set_gain (src/gr/gr_demod_base.cpp:1155-1185, 1206-1210): the analogue receivers' own setters
@pytest.mark.parametrize("kind,modem,fw,w", [("nbfm", 9, 5000, 4000), ("nbfm", 8, 2500, 3000), ("am", 14, 5000, 4000), ("wbfm", 10, 75000, 60000)])
@pytest.mark.parametrize("chunk", [1 << 20, 33334])
def test_analog_set_filter_width_bit_exact(qrl_ctx, kind, modem, fw, w, chunk):
    """qrl_demod_set_filter_width against the oracle's chain with the reference setter's designs (pinned by tests/test_ref_chains.py::
    test_set_filter_width_of_the_analogue_blocks): low_pass(1, fs, w, 1200, BH) + new discriminator gain (NBFM, WBFM), complex_band_pass (AM)"""
    import torch
    import qradiolink_amd as q
    n = 400000
    xs = [sig.make_analog(kind, n=n, seed=1, gap=(100000, 300000))[0], sig.make_analog(kind, n=n, seed=2)[0]]
    iq = np.stack(xs)
    dem = q.Demod(qrl_ctx, modem, batch=2, max_chunk=min(chunk, n))
    q.collect(dem, torch.from_numpy(iq[:, :50000].copy()).cuda(), min(chunk, 50000))        # something in flight: the setter restarts the chain
    dem.set_filter_width(w)
    out = q.collect(dem, torch.from_numpy(iq).cuda(), min(chunk, n))
    with pytest.raises(q.QrlError):
        dem.set_filter_width(0)
    dem.close()
    for b in range(2):
        ref = orc.demod_analog(iq[b], kind, filter_width=fw, set_width=w)
        assert ref["audio"].size > 1500
        got, want = out["filtered"][b].view(np.float32) + np.float32(0), ref["filtered"].view(np.float32) + np.float32(0)
        assert got.size == want.size and np.array_equal(got.view(np.uint32), want.view(np.uint32)), "filtered"
        got, want = out["audio"][b] + np.float32(0), ref["audio"] + np.float32(0)
        assert got.size == want.size and np.array_equal(got.view(np.uint32), want.view(np.uint32)), "audio"
        ctor = orc.demod_analog(iq[b], kind, filter_width=w)                                  # not the constructor's chain with that width
        assert not np.array_equal(ctor["filtered"], ref["filtered"])


@pytest.mark.parametrize("lsb,modem", [(False, 11), (True, 12)])
def test_ssb_set_filter_width_and_set_gain_bit_exact(qrl_ctx, lsb, modem):
    """gr_demod_ssb::set_filter_width (the band-pass with the new edge, audio filter with GAIN 2) and set_gain (_if_gain) against the oracle"""
    import torch
    import qradiolink_amd as q
    n = 1200000
    iq = np.stack([sig.make_ssb(n=n, seed=1, lsb=lsb, gap=(300000, 900000)), sig.make_ssb(n=n, seed=2, lsb=lsb)])
    dem = q.Demod(qrl_ctx, modem, batch=2, max_chunk=150000)
    dem.set_filter_width(2400)
    dem.set_gain(0.5)
    out = q.collect(dem, torch.from_numpy(iq).cuda(), 150000)
    for b in range(2):
        ref = orc.demod_ssb(iq[b], sb=int(lsb), set_width=2400, gain=0.5)
        assert ref["audio"].size >= 4096
        got, want = out["filtered"][b].view(np.float32) + np.float32(0), ref["filtered"].view(np.float32) + np.float32(0)
        assert got.size == want.size and np.array_equal(got.view(np.uint32), want.view(np.uint32)), "filtered"
        got, want = out["audio"][b] + np.float32(0), ref["audio"] + np.float32(0)
        assert got.size == want.size and np.array_equal(got.view(np.uint32), want.view(np.uint32)), "audio"
    # set_gain alone is live (no restart): the next call's port 0 scales, the stream position runs on
    dem.reset()
    dem.set_gain(0.9)
    a = q.collect(dem, torch.from_numpy(iq[:, :300000].copy()).cuda(), 150000)
    ref = orc.demod_ssb(iq[0][:300000], sb=int(lsb), set_width=2400)
    assert np.array_equal(a["filtered"][0], ref["filtered"])
    with pytest.raises(q.QrlError):
        dem.set_filter_width(150)                                                           # below the 200 Hz band edge
    dem.close()
    d2 = q.Demod(qrl_ctx, q.MODEM_NBFM5000, batch=1, max_chunk=1000)
    with pytest.raises(q.QrlError):
        d2.set_gain(0.5)                                                                    # SSB receivers only
    d2.close()
    d3 = q.Demod(qrl_ctx, 0, batch=1, max_chunk=1000)
    with pytest.raises(q.QrlError):
        d3.set_filter_width(3000)                                                           # analogue receivers only
    d3.close()
