"""Second, INDEPENDENT restatement of the recursive GNU Radio blocks of the RX chains, written from the block semantics in
SURVEY.md Appendix A (A.5 control_loop / fll_band_edge_cc, A.6 symbol_sync_ff/cc, A.7 MMSE interpolator, A.9 cc_decoder,
A.10 LFSR, A.11 agc2_cc, A.12 costas_loop_cc) in plain Python / numpy with DOUBLE-precision arithmetic and the textbook loop
structure -- NOT from oracle/*.c, which states the same blocks in float32 with GPU-shaped summation orders.

Purpose (tests/test_independent_restatement.py): two restatements that were written separately must agree -- decisions
identical, floats within float32-vs-float64 loop noise -- or one of them carries a transcription error.  This does not pin
either to GNU Radio itself (both come from the same recollection of upstream; parity stays "unpinned").

Where Appendix A and this file differ on purpose:
  * A.6 writes the real-valued modified M&M error as clip(u, 1) / 2; upstream timing_error_detector.cc computes
    branchless_clip(u / 2, 1).  The latter is used here (and in the oracle)."""
import math

import numpy as np

TWO_PI = 2.0 * math.pi


# ---------------------------------------------------------------- A.5 control_loop, fll_band_edge_cc
def control_loop_gains(bw):
    zeta = math.sqrt(2.0) / 2.0
    denom = 1.0 + 2.0 * zeta * bw + bw * bw
    return 4.0 * zeta * bw / denom, 4.0 * bw * bw / denom


def _sinc(x):
    return 1.0 if x == 0 else math.sin(math.pi * x) / (math.pi * x)


def fll_taps(sps, rolloff, n):
    """design_filter of fll_band_edge_cc: returns (lower, upper) as handed to the two FIR filters"""
    m = round(n / sps)
    bb = [_sinc(rolloff * (-m + i * 2.0 / sps) - 0.5) + _sinc(rolloff * (-m + i * 2.0 / sps) + 0.5) for i in range(n)]
    power = sum(bb)
    nc = int((n - 1.0) / 2.0)
    lower, upper = [0j] * n, [0j] * n
    for i in range(n):
        t = bb[i] / power
        k = (-nc + i) / (2.0 * sps)
        lower[n - 1 - i] = t * np.exp(-1j * TWO_PI * (1 + rolloff) * k)
        upper[n - 1 - i] = t * np.exp(+1j * TWO_PI * (1 + rolloff) * k)
    return np.array(lower), np.array(upper)


def _phase_wrap(p):
    while p > TWO_PI:
        p -= TWO_PI
    while p < -TWO_PI:
        p += TWO_PI
    return p


def fll_band_edge(x, sps, rolloff, ntaps, bw, delay=True):
    """out[i] = in[i] * e^{j phase}; the band-edge filters run over the last ntaps OUTPUTS (fir_filter_with_buffer: the filter
    holds the REVERSED tap vector, i.e. stored tap j multiplies the output that is ntaps-1-j samples old); error = |l|^2 - |u|^2.
    delay: history() = ntaps + 1 shifts the stream by ntaps samples (the first ntaps outputs see zero input)."""
    lower, upper = fll_taps(sps, rolloff, ntaps)
    alpha, beta = control_loop_gains(bw)
    fmax = TWO_PI * (2.0 / sps)
    phase = freq = 0.0
    line = np.zeros(ntaps, complex)    # line[j] = output j samples ago
    out = np.zeros(len(x), complex)
    for i in range(len(x)):
        xi = (x[i - ntaps] if i >= ntaps else 0.0) if delay else x[i]
        y = xi * complex(math.cos(phase), math.sin(phase))
        out[i] = y
        line[1:] = line[:-1]
        line[0] = y
        u = np.dot(upper[::-1], line)
        lo = np.dot(lower[::-1], line)
        err = abs(lo) ** 2 - abs(u) ** 2
        freq += beta * err
        phase += freq + alpha * err
        phase = _phase_wrap(phase)
        freq = min(max(freq, -fmax), fmax)
    return out


# ---------------------------------------------------------------- A.12 costas_loop_cc
def tanh_lut():
    return np.array([math.tanh((i - 128) / 64.0) for i in range(256)])


def _tanhf_lut(x, table):
    if x > 2.0:
        return 1.0
    if x <= -2.0:
        return -1.0
    return table[min(max(int(128.0 + 64.0 * x), 0), 255)]


def _clip(x, c):
    return 0.5 * (abs(x + c) - abs(x - c))


def costas(x, bw, order, use_snr):
    alpha, beta = control_loop_gains(bw)
    table = tanh_lut()
    phase = freq = 0.0
    out = np.zeros(len(x), complex)
    for i in range(len(x)):
        y = x[i] * complex(math.cos(-phase), math.sin(-phase))
        out[i] = y
        if order == 2:
            e = y.real * y.imag
        elif use_snr:
            snr = abs(y) ** 2
            e = _tanhf_lut(snr * y.real, table) * y.imag - _tanhf_lut(snr * y.imag, table) * y.real
        else:
            e = (1.0 if y.real > 0 else -1.0) * y.imag - (1.0 if y.imag > 0 else -1.0) * y.real
        e = _clip(e, 1.0)
        freq += beta * e
        phase += freq + alpha * e
        phase = _phase_wrap(phase)
        freq = min(max(freq, -1.0), 1.0)
    return out


# ---------------------------------------------------------------- A.11 agc2_cc
def agc2(x, attack, decay, ref, gain, max_gain=65536.0):
    out = np.zeros(len(x), complex)
    g = gain
    for i in range(len(x)):
        out[i] = x[i] * g
        tmp = abs(out[i]) - ref
        rate = attack if tmp > g else decay
        g -= tmp * rate
        if g < 0:
            g = 10e-5
        if max_gain > 0 and g > max_gain:
            g = max_gain
    return out


# ---------------------------------------------------------------- A.7 MMSE interpolator table (closed form, 6 significant digits)
def mmse_table():
    B = 0.25
    t = np.arange(8) - 3.0
    R = np.sinc(2 * B * (t[:, None] - t[None, :]))
    rows = np.zeros((129, 8))
    for imu in range(129):
        mu = imu / 128.0
        c = np.linalg.solve(R, np.sinc(2 * B * (t - mu)))
        rows[imu] = [float("%.5e" % v) for v in c[::-1]]
    rows[0] = [0, 0, 0, 0, 1, 0, 0, 0]
    rows[128] = [0, 0, 0, 1, 0, 0, 0, 0]
    return rows


# ---------------------------------------------------------------- A.6 symbol_sync_ff / symbol_sync_cc
def clock_loop_gains(loop_bw, zeta, ted_gain):
    omega_n_t = loop_bw          # loop_bw is the normalised natural radian frequency
    k0 = 2.0 / ted_gain
    k1 = math.exp(-zeta * omega_n_t)
    sh = math.sinh(zeta * omega_n_t)
    if zeta > 1.0:
        cx = math.cosh(omega_n_t * math.sqrt(zeta * zeta - 1.0))
    elif zeta == 1.0:
        cx = 1.0
    else:
        cx = math.cos(omega_n_t * math.sqrt(1.0 - zeta * zeta))
    return k0 * k1 * sh, k0 * (1.0 - k1 * (sh + cx))


def _slice(constellation, v):
    if constellation == "bpsk":
        return complex(1.0 if v.real >= 0 else -1.0, 0.0)
    if constellation == "dqpsk":
        s = math.sqrt(0.5)
        return complex(s if v.real >= 0 else -s, s if v.imag >= 0 else -s)
    if constellation == "4level":   # constellation_rect {-1.5, -0.5, 0.5, 1.5}: sector (int)(x + 2) clamped to [0, 3]
        k = min(max(int(math.floor(v.real + 2.0)), 0), 3)
        return complex(-1.5 + k, 0.0)
    raise ValueError(constellation)


def _modmm(u, variant):
    """include/qrl_contracts.h: 0 clip(u/2, 1) | 1 clip(u, 1)/2 | 2 clip(u, 1)"""
    return _clip(u / 2.0, 1.0) if variant == 0 else (_clip(u, 1.0) / 2.0 if variant == 1 else _clip(u, 1.0))


def symbol_sync(x, ted, sps, loop_bw, damping, ted_gain, max_dev, constellation, complex_in, modmm=None):
    """ted: 'mm' | 'mod_mm'.  Returns the interpolated symbols (osps = 1, MMSE 8-tap).
    modmm: candidate formula of the modified M&M error (None = the contract: 0 for real input, 2 for complex input)."""
    if modmm is None:
        modmm = 2 if complex_in else 0
    taps = mmse_table()
    alpha, beta = clock_loop_gains(loop_bw, damping, ted_gain)
    avg = inst = float(sps)
    pmax, pmin = sps + max_dev, sps - max_dev
    mu, ii = 0.0, 0
    xs = [0j, 0j, 0j]      # newest first
    ds = [0j, 0j, 0j]
    out = []
    x = np.asarray(x)
    while ii + 8 <= len(x):
        imu = int(round(mu * 128.0))
        y = complex(np.dot(taps[imu][::-1], x[ii:ii + 8]))
        out.append(y)
        xs = [y, xs[0], xs[1]]
        ds = [_slice(constellation, y), ds[0], ds[1]]
        if ted == "mm":
            if complex_in:
                e = (ds[1].real * xs[0].real - ds[0].real * xs[1].real) + (ds[1].imag * xs[0].imag - ds[0].imag * xs[1].imag)
            else:
                e = ds[1].real * xs[0].real - ds[0].real * xs[1].real
        else:
            if complex_in:
                u = (xs[0] - xs[2]) * ds[1].conjugate() - (ds[0] - ds[2]) * xs[1].conjugate()
                e = _modmm(u.real, modmm)
            else:
                u = (xs[0].real - xs[2].real) * ds[1].real - (ds[0].real - ds[2].real) * xs[1].real
                e = _modmm(u, modmm)
        avg += beta * e
        avg = min(max(avg, pmin), pmax)
        inst = avg + alpha * e
        if inst <= 0:
            inst = avg
        ph = mu + inst
        n = math.floor(ph)
        mu = ph - n
        ii += int(n)
    return np.array(out)


# ---------------------------------------------------------------- A.9 cc_decoder (K = 7, rate 1/2, streaming, frame 80)
def _parity(v):
    return bin(v).count("1") & 1


def cc_decode_k7(soft, frame=80, variant="generic"):
    """cc_decoder(frame 80, K 7, rate 1/2, CC_STREAMING): per frame 80 + 6 look-ahead trellis steps from the start state (all
    metrics 63, start state 0), chainback from the best end state (first minimum); the decision of step t + 6 is bit t; the
    state reached 6 steps before the end seeds the next frame.
    variant "generic": volk_8u_x4_conv_k7_r2_8u_generic as written in A.9 -- metric = ((B0^s0)>>1 + (B1^s1)>>1) >> 2, sums in
        unsigned int, decision = (int)(m0 - m1) > 0 (a tie keeps the lower predecessor), survivors stored as unsigned char,
        minimum subtracted after every step.
    variant "spiral": the SSE kernel x86 VOLK actually dispatches to (and the reference documents as the working one,
        docs/OPERATION.md:4) -- metric = (avg_epu8(B0^s0, B1^s1) >> 2) & 63 (avg rounds up), saturating adds, survivor = unsigned
        minimum with the decision bit SET on a tie, minimum subtracted only when metric[0] > 210."""
    soft = np.asarray(soft, np.uint8).astype(np.int64)
    polys = (109, 79)
    b0 = np.array([255 if _parity((2 * s) & polys[0]) else 0 for s in range(32)], np.int64)
    b1 = np.array([255 if _parity((2 * s) & polys[1]) else 0 for s in range(32)], np.int64)
    nframes = max((len(soft) - 12) // (2 * frame), 0)
    bits = []
    start = 0
    for f in range(nframes):
        sym = soft[2 * frame * f: 2 * frame * f + 2 * (frame + 6)]
        X = np.full(64, 63, np.int64)
        X[start] = 0
        dec = np.zeros((frame + 6, 64), np.int64)
        for t in range(frame + 6):
            a, b = b0 ^ sym[2 * t], b1 ^ sym[2 * t + 1]
            if variant == "generic":
                m = ((a >> 1) + (b >> 1)) >> 2
                m0, m1, m2, m3 = X[:32] + m, X[32:] + (63 - m), X[:32] + (63 - m), X[32:] + m
                d0, d1 = (m0 - m1) > 0, (m2 - m3) > 0
                Y = np.empty(64, np.int64)
                Y[0::2] = np.where(d0, m1, m0) & 0xFF
                Y[1::2] = np.where(d1, m3, m2) & 0xFF
                Y = Y - Y.min()
            else:
                m = (((a + b + 1) >> 1) >> 2) & 63
                m0, m1 = np.minimum(X[:32] + m, 255), np.minimum(X[32:] + (63 - m), 255)
                m2, m3 = np.minimum(X[:32] + (63 - m), 255), np.minimum(X[32:] + m, 255)
                d0, d1 = m1 <= m0, m3 <= m2
                Y = np.empty(64, np.int64)
                Y[0::2] = np.minimum(m0, m1)
                Y[1::2] = np.minimum(m2, m3)
                if Y[0] > 210:
                    Y = Y - Y.min()
            dec[t, 0::2] = d0
            dec[t, 1::2] = d1
            X = Y
        state = int(np.argmin(X))
        out = [0] * frame
        for nb in range(frame - 1, -1, -1):
            k = int(dec[nb + 6, state])
            state = (state >> 1) | (k << 5)
            out[nb] = k
            if nb == frame - 6:
                start = state
        bits.extend(out)
    return np.array(bits, np.uint8)


# ---------------------------------------------------------------- A.10 LFSR descrambler
def descramble(bits, mask=0x8A, seed=0x7F, length=7):
    sr = seed
    out = []
    for b in bits:
        b = int(b) & 1
        out.append(_parity(sr & mask) ^ b)
        sr = (sr >> 1) | (b << length)
    return np.array(out, np.uint8)
