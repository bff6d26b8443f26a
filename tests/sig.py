"""Synthetic IQ for tests and bench (TEST INFRASTRUCTURE): frames built like gr_modem::frame()
(reference src/gr_modem.cpp:904-961), modulated by the oracle's restatement of the reference
modulators, passed through a seeded channel and optionally interpolated to the SDR rate like
gr_mod_base (gr_mod_base.cpp:249-250) with the RX tuning offset applied, so the demodulator's
rotator has work to do.

Two channels:
* the DEFAULT one (what the golden fixtures were made with): CFO + AWGN + an INTEGER lead of zeros;
* SURVEY 8(d)'s channel, `impair=SPEC` (or any `Impair(...)`): the modulator output is RESAMPLED
  first -- a fractional delay (0.37 sample) and a clock error (+20 ppm: the transmitter's sample
  clock runs fast, so the receiver sees a slowly sliding symbol phase) by a 32-tap Kaiser-windowed
  sinc evaluated per output sample in float64 -- then CFO, then AWGN at a stated Es/N0 (12 dB;
  Es = signal power x samples per channel symbol, N0 = complex noise variance per sample).
  This is what exercises the timing loops' stuff / skip and clamp branches (symbol_sync_ff max_dev
  0.1 gr_demod_2fsk.cpp:106-110, symbol_sync_cc max_dev 8e-4 gr_demod_qpsk.cpp:105-109,
  clock_recovery_mm_cc gr_demod_bpsk.cpp:51-103) systematically, also across call cuts."""
import numpy as np

import orc

MODES = {
    # name: (modulator, kwargs, sync bytes, payload bytes, nominal cfo, snr_db at 1 Msps)
    "2fsk1k": (orc.mod_2fsk, dict(sps=50, filter_width=2000, fm=False), bytes([0xB5]), 4, 137.0, -8.0),
    "2fsk1kfm": (orc.mod_2fsk, dict(sps=50, filter_width=2500, fm=True), bytes([0xB5]), 4, 137.0, -8.0),
    "gmsk10k": (orc.mod_gmsk, dict(sps=10, filter_width=20000), bytes([0xED, 0x89, 0xAA]), 47, 137.0, 4.0),
    "gmsk1k": (orc.mod_gmsk, dict(sps=100, filter_width=2000), bytes([0xB5]), 4, 137.0, -8.0),
    "qpsk250k": (orc.mod_qpsk, dict(sps=4, filter_width=160000), bytes([0xDE, 0x98, 0xAA]), 1516, 1300.0, 12.0),
    "qpsk2k": (orc.mod_qpsk, dict(sps=500, filter_width=1300), bytes([0xED, 0x89, 0xAA]), 7, 5.0, -6.0),
    "qpsk20k": (orc.mod_qpsk, dict(sps=100, filter_width=6500), bytes([0xED, 0x89, 0xAA]), 47, 20.0, 2.0),
    # native 4FSK (FM variants; no FLL in gr_demod_4fsk, so only a small CFO) and BPSK
    "4fsk2k": (orc.mod_4fsk, dict(sps=25, filter_width=4000, fm=False), bytes([0xED, 0x89, 0xAA]), 7, 15.0, 0.0),
    "4fsk2kfm": (orc.mod_4fsk, dict(sps=25, filter_width=3500, fm=True), bytes([0xED, 0x89, 0xAA]), 7, 15.0, 0.0),
    "4fsk1kfm": (orc.mod_4fsk, dict(sps=50, filter_width=2000, fm=True), bytes([0xB5]), 4, 10.0, 0.0),
    "4fsk10kfm": (orc.mod_4fsk, dict(sps=5, filter_width=20000, fm=True), bytes([0xED, 0x89, 0xAA]), 47, 40.0, 8.0),
    "4fsk100k": (orc.mod_4fsk, dict(sps=2, filter_width=125000, fm=True), bytes([0xDE, 0x98, 0xAA]), 1516, 300.0, 16.0),
    "bpsk1k": (orc.mod_bpsk, dict(sps=500, filter_width=1500), bytes([0xB5]), 4, 3.0, -6.0),
    "bpsk2k": (orc.mod_bpsk, dict(sps=250, filter_width=2800), bytes([0xED, 0x89, 0xAA]), 7, 3.0, -4.0),
}


def frames(mode, nframes, rng):
    sync, plen = MODES[mode][2], MODES[mode][3]
    payloads = [bytes(rng.integers(0, 256, plen, dtype=np.uint8)) for _ in range(nframes)]
    data = bytes([0xAA] * 8) + b"".join(sync + p for p in payloads) + bytes([0xAA] * 8)
    return np.frombuffer(data, np.uint8), payloads


class Impair:
    """channel impairments beyond CFO / AWGN: fractional delay in samples, clock error in ppm (positive: the transmitter's clock is
    fast, the received waveform is compressed), Es/N0 in dB (None: the mode's per-sample SNR of MODES)"""

    def __init__(self, frac_delay=0.0, clock_ppm=0.0, esn0_db=None, name=None):
        self.frac_delay, self.clock_ppm, self.esn0_db = float(frac_delay), float(clock_ppm), esn0_db
        self.name = name or "delay%.2f_ppm%g_esn0%s" % (frac_delay, clock_ppm, "mode" if esn0_db is None else "%g" % esn0_db)

    def __repr__(self):
        return self.name


SPEC = Impair(0.37, 20.0, 12.0, name="survey8d")            # 0.37 sample, +20 ppm, Es/N0 12 dB
SPEC_CLEAN = Impair(0.37, 20.0, None, name="survey8d_modesnr")   # same timing, the mode's own SNR
# Clock errors large enough to drive the timing loops into their limiters -- each one MEASURED with the oracle's limiter-hit counter
# (orc.loop_clamp_hits, tests/test_channel_8d.py): whether a loop reaches its clamp depends on its bandwidth and on the stream's length
# (symbol_sync_cc of QPSK-250k: max_dev 8e-4 sample behind a loop bandwidth of 2 pi / 25000 needs > 1e5 symbols; clock_recovery_mm_cc of the
# BPSK chains: gain_omega 2.5e-5 against a limit of 1e-3 omega needs thousands), so the table carries the frame count too.
# (sig mode, clock error in ppm, frames per stream, minimum limiter hits of one stream in the oracle)
CLAMP_CASES = {
    "2fsk1k": (15000.0, 2, 100), "gmsk10k": (-15000.0, 2, 300), "4fsk2kfm": (-5000.0, 2, 5), "4fsk100k": (-15000.0, 2, 10000),
    "4fsk2k": (5000.0, 2, 10), "qpsk20k": (15000.0, 2, 30), "qpsk250k": (1000.0, 12, 50000), "bpsk2k": (3000.0, 30, 500),
    "bpsk1k": (3000.0, 30, 200),
}


def clamp_impair(mode):
    ppm, nframes, _ = CLAMP_CASES[mode]
    return Impair(0.37, ppm, None, name="drift%+gppm" % ppm), nframes


def resample_clock(x, frac_delay, clock_ppm, half=16, beta=8.0):
    """y[k] = x((k - frac_delay) * (1 + clock_ppm * 1e-6)): band-limited interpolation with a Kaiser-windowed sinc of 2 * half taps,
    float64, evaluated per output sample (the test signals occupy a small part of the band, so 32 taps are far below the noise)"""
    if frac_delay == 0.0 and clock_ppm == 0.0:
        return np.asarray(x, np.complex128)
    x = np.asarray(x, np.complex128)
    r = 1.0 + clock_ppm * 1e-6
    nout = int(np.floor((x.size - 1) / r + frac_delay))
    t = (np.arange(nout, dtype=np.float64) - frac_delay) * r
    i0 = np.floor(t).astype(np.int64)
    mu = t - i0
    xp = np.concatenate([np.zeros(half, np.complex128), x, np.zeros(half + 1, np.complex128)])
    y = np.zeros(nout, np.complex128)
    wsum = np.zeros(nout, np.float64)
    from scipy.special import i0 as bessel_i0
    for k in range(-half + 1, half + 1):
        d = k - mu                                        # distance of tap k from the interpolation point, in samples
        w = np.sinc(d) * bessel_i0(beta * np.sqrt(np.clip(1.0 - (d / half) ** 2, 0.0, 1.0)))
        wsum = wsum + w
        y += w * xp[i0 + k + half]
    return y / wsum                                       # unit gain at DC for every fractional offset


def channel(x, fs, cfo, snr_db, amp, rng, lead=0, impair=None, samples_per_symbol=None):
    if impair is not None:
        x = resample_clock(x, impair.frac_delay, impair.clock_ppm)
        if impair.esn0_db is not None:
            snr_db = impair.esn0_db - 10.0 * np.log10(samples_per_symbol)
    n = np.arange(x.size + lead)
    y = np.zeros(x.size + lead, np.complex128)
    y[lead:] = amp * x
    y *= np.exp(2j * np.pi * cfo * n / fs)
    p = np.mean(np.abs(amp * x) ** 2)
    sigma = np.sqrt(p / 10 ** (snr_db / 10) / 2)
    y += sigma * (rng.standard_normal(y.size) + 1j * rng.standard_normal(y.size))
    return y.astype(np.complex64)


def samples_per_symbol(mode, nsamples, nbytes):
    """samples per CHANNEL symbol at 1 Msps: every chain is K = 7 rate 1/2 coded; QPSK and 4FSK carry two coded bits per symbol"""
    bps = 2 if (mode.startswith("qpsk") or mode.startswith("4fsk")) else 1
    return nsamples / (8.0 * nbytes) * bps / 2.0


def make_stream(mode, nframes=4, device_rate=1000000, rx_offset_hz=25000.0, seed=1, amp=0.05, lead=0, impair=None):
    """One stream at device_rate.  Returns (iq complex64, payload list).  impair: None = the default channel, or an Impair."""
    rng = np.random.default_rng(seed)
    mod, kw, _, _, cfo, snr = MODES[mode]
    data, payloads = frames(mode, nframes, rng)
    x = mod(data, **kw)
    y = channel(x, 1e6, cfo + 13.0 * (seed % 7), snr, amp, rng, lead=lead, impair=impair,
                samples_per_symbol=samples_per_symbol(mode, x.size, data.size))
    if device_rate >= 2000000:
        y = orc.tx_interp(y, device_rate)
        n = np.arange(y.size)
        # the signal sits rx_offset_hz above the tuned centre; the RX rotator (-offset) brings it back
        y = (y * np.exp(2j * np.pi * rx_offset_hz * n / device_rate)).astype(np.complex64)
    return y, payloads


def make_batch(mode, batch, nframes=4, device_rate=1000000, rx_offset_hz=25000.0, seed=1, amp=0.05, impair=None):
    streams = [make_stream(mode, nframes, device_rate, rx_offset_hz, seed + 101 * b, amp, lead=37 * b, impair=impair)[0] for b in range(batch)]
    n = min(s.size for s in streams) & ~1
    return np.stack([s[:n] for s in streams]).astype(np.complex64)


def find_frames(bits, sync, nbits):
    """gr_modem::findSync-style shift-register search (src/gr_modem.cpp:1183-1282), returns payloads."""
    out = []
    nsync = 8 * len(sync)
    want = int.from_bytes(sync, "big")
    mask = (1 << nsync) - 1
    sr = 0
    i, n = 0, len(bits)
    while i < n:
        sr = ((sr << 1) | int(bits[i])) & mask
        i += 1
        if sr == want and i + nbits <= n:
            out.append(np.packbits(bits[i:i + nbits]).tobytes())
            i += nbits
            sr = 0
    return out


def make_4fsk(nsym=400, seed=1, amp=0.3, noise=0.002, cfo=0.0, fs=1000000.0, levels=None, alpha=0.2, dev=1944.0, clock_ppm=0.0, frac_delay=0.0):
    """DMR-like 4FSK at 4800 sym/s on 1 Msps IQ (RRC alpha 0.2, deviation +-1944 / +-648 Hz; dibit map of the DMR air
    interface: 01 -> +3, 00 -> +1, 10 -> -1, 11 -> -3).  Returns (iq complex64, dibits).  levels: explicit symbol levels in
    units of the outer deviation (+-1, +-1/3, 0 = unmodulated carrier) instead of random dibits.  M17: alpha=0.5, dev=2400.
    clock_ppm: the transmitter's symbol clock runs fast by that much (symbol rate 4800 (1 + ppm 1e-6)); frac_delay: the first symbol sits that many
    SYMBOL periods late (SURVEY 8(d)'s timing impairments for the 4FSK tails, where the signal is synthesised at the sample rate directly)."""
    rng = np.random.default_rng(seed)
    dib = rng.integers(0, 4, nsym)
    lev = np.array([+1, +3, -1, -3])[dib] / 3.0
    if levels is not None:
        lev = np.asarray(levels, float)
        nsym, dib = lev.size, None
    sps = fs / (4800.0 * (1.0 + clock_ppm * 1e-6))
    n = int((nsym + frac_delay) * sps) & ~1
    up = np.zeros(n)
    pos = ((np.arange(nsym) + frac_delay) * sps).astype(int)
    up[pos[pos < n]] = lev[:np.count_nonzero(pos < n)]
    L = int(8 * sps)
    tt = np.arange(-L, L + 1) / sps
    a = alpha
    with np.errstate(divide="ignore", invalid="ignore"):
        h = (np.sin(np.pi * tt * (1 - a)) + 4 * a * tt * np.cos(np.pi * tt * (1 + a))) / (np.pi * tt * (1 - (4 * a * tt) ** 2))
    h[np.isnan(h)] = 1 - a + 4 * a / np.pi
    h[np.isinf(h)] = 0
    if float(n) * h.size > 2e8:     # long streams at a high sample rate (bench.py's C4 input): the same pulse shaping through an FFT
        from scipy.signal import fftconvolve
        f = fftconvolve(up, h, mode="same")
    else:
        f = np.convolve(up, h, mode="same")
    ph = 2 * np.pi * np.cumsum(f * dev + cfo) / fs
    x = amp * np.exp(1j * ph) + noise * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    return x.astype(np.complex64), dib


# ---------------------------------------------------------------- DMR bursts for the DMO correlator slicer (a37b)
DMR_MS_DATA_SYNC, DMR_MS_VOICE_SYNC = 0xD5D7F77FD757, 0x7F7D5DD57DFD   # reference src/DMR/constants.h:8-9


def golay2087_encode(v):
    """(20,8) slot-type code word of byte v as 20 bits: the (19,8) cyclic code word (generator 0xC75) + overall parity"""
    pattern = v << 11
    aux = 0x40000
    p = pattern
    while p & 0xFFFFF800:
        while not (aux & p):
            aux >>= 1
        p ^= (aux // 0x800) * 0xC75
    cw19 = pattern | p
    return (cw19 << 1) | (bin(cw19).count("1") & 1)


def dmr_frame(info_bits, sync=None, colour_code=1, data_type=None):
    """264-bit DMR burst as 33 bytes: info[0:98] | slot type[0:10] | sync (48) | slot type[10:20] | info[98:196]; without
    sync / slot type (voice frames B-F) the middle 68 bits come from `info_bits` as well (it must then hold 264 bits)."""
    info_bits = np.asarray(info_bits, np.uint8)
    if sync is None:
        bits = info_bits[:264].copy()
    else:
        st = [(golay2087_encode(((colour_code & 15) << 4) | (data_type & 15)) >> (19 - i)) & 1 for i in range(20)] if data_type is not None \
            else list(info_bits[196:216])
        sy = [(sync >> (47 - i)) & 1 for i in range(48)]
        bits = np.concatenate([info_bits[:98], st[:10], sy, st[10:], info_bits[98:196]]).astype(np.uint8)
    return np.packbits(bits).tobytes()


def dmr_samples(frames, gap=780, scale=0.3, sps=5, lead=300):
    """discriminator-like float stream at 24 ksps: dibit 01 -> +3, 00 -> +1, 10 -> -1, 11 -> -3 (x scale / 3), `sps` samples per
    symbol, `gap` silent samples between bursts (DMO: one 27.5 ms burst every 60 ms = 1440 samples)"""
    lv = {(0, 1): 3.0, (0, 0): 1.0, (1, 0): -1.0, (1, 1): -3.0}
    out = [np.zeros(lead, np.float32)]
    for f in frames:
        bits = np.unpackbits(np.frombuffer(f, np.uint8))
        sym = np.array([lv[(int(bits[2 * i]), int(bits[2 * i + 1]))] for i in range(132)], np.float32) * (scale / 3.0)
        out.append(np.repeat(sym, sps))
        out.append(np.zeros(gap, np.float32))
    return np.concatenate(out)


def dmr_levels(frames, gap_symbols=156, lead_symbols=60):
    """symbol levels (units of the outer deviation) of DMO bursts separated by unmodulated carrier: 132 symbols per burst, one
    burst per 288 symbols = 60 ms"""
    lv = {(0, 1): 1.0, (0, 0): 1.0 / 3, (1, 0): -1.0 / 3, (1, 1): -1.0}
    out = [np.zeros(lead_symbols)]
    for f in frames:
        bits = np.unpackbits(np.frombuffer(f, np.uint8))
        out.append(np.array([lv[(int(bits[2 * i]), int(bits[2 * i + 1]))] for i in range(132)]))
        out.append(np.zeros(gap_symbols))
    return np.concatenate(out)


def make_dsss(info_bits, seed=1, amp=0.05, noise=0.0005, cfo=0.0):
    """DSSS "BPSK 8" burst at 1 Msps, built like gr_mod_dsss (src/gr/gr_mod_dsss.cpp:33-92): scrambler -> K=7 CC -> Barker-13
    spreading (bit 0 -> code, bit 1 -> inverted code) -> {-1, +1} -> RRC at 25 samples per chip (5200 sps) -> 0.65 -> up to
    1 Msps (exact ratio 2500 / 13, scipy polyphase instead of the reference's two rational resamplers: a test signal)."""
    from scipy.signal import resample_poly
    rng = np.random.default_rng(seed)
    enc = orc.cc_encode_k7(orc.scramble(np.asarray(info_bits, np.uint8)))
    code = np.array([1, 1, 1, 1, 1, 0, 0, 1, 1, 0, 1, 0, 1], np.uint8)
    chips = np.concatenate([code if b == 0 else 1 - code for b in enc])
    sps = 25
    rrc = orc.root_raised_cosine(sps, sps, 1, 0.35, 11 * sps).astype(np.float64)
    up = np.zeros(chips.size * sps)
    up[::sps] = chips.astype(np.float64) * 2 - 1
    x = np.convolve(up, rrc)[:up.size] * 0.65
    y = resample_poly(x, 2500, 13)
    n = np.arange(y.size)
    z = amp * y * np.exp(2j * np.pi * cfo * n / 1e6) + noise * (rng.standard_normal(y.size) + 1j * rng.standard_normal(y.size))
    return z.astype(np.complex64)


def make_analog(kind, n=400000, seed=1, amp=0.05, noise=0.0005, gap=None, fs=1000000.0):
    """Analogue voice test signal at 1 Msps: two audio tones, FM (deviation 2.5 kHz / 50 kHz for WBFM) or AM (60 % depth) on the
    carrier at 0 Hz, AWGN; `gap` = (start, stop) zeroes that span completely (an idle channel: the gating squelch closes)."""
    rng = np.random.default_rng(seed)
    t = np.arange(n) / fs
    audio = 0.6 * np.sin(2 * np.pi * (700.0 + 13 * seed) * t) + 0.3 * np.sin(2 * np.pi * (1900.0 + 7 * seed) * t + 0.4)
    if kind == "am":
        x = (1.0 + 0.6 * audio) * np.exp(0.3j)
    else:
        dev = 50000.0 if kind == "wbfm" else 2500.0
        x = np.exp(2j * np.pi * dev * np.cumsum(audio) / fs)
    y = amp * x + noise * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    if gap:
        y[gap[0]:gap[1]] = 0
    return y.astype(np.complex64), audio


def make_ssb(n=400000, seed=1, amp=0.05, noise=0.0003, lsb=False, gap=None, fs=1000000.0):
    """SSB test signal at 1 Msps: two audio tones as complex exponentials above (USB) or below (LSB) the suppressed carrier at 0 Hz"""
    rng = np.random.default_rng(seed)
    t = np.arange(n) / fs
    sgn = -1.0 if lsb else 1.0
    x = 0.6 * np.exp(sgn * 2j * np.pi * (700.0 + 13 * seed) * t) + 0.3 * np.exp(sgn * 2j * np.pi * (1900.0 + 7 * seed) * t + 0.4j)
    y = amp * x + noise * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    if gap:
        y[gap[0]:gap[1]] = 0
    return y.astype(np.complex64)
