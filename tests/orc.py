"""ctypes binding of the CPU oracle (oracle/liborc.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the
product package."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORC_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = os.path.join(_ORC_DIR, "liborc.so")

cf32 = np.complex64


def build():
    subprocess.check_call(["make", "-s", "-C", _ORC_DIR, "liborc.so"])


def _load():
    if not os.path.exists(_LIB):
        build()
    return C.CDLL(_LIB)


lib = _load()

_p = C.c_void_p
_sz = C.c_size_t


class DemodOut(C.Structure):
    _fields_ = [("n_filtered", _sz), ("n_const", _sz), ("n_bits_a", _sz), ("n_bits_b", _sz),
                ("filtered", _p), ("constellation", _p), ("bits_a", _p), ("bits_b", _p)]


def _ptr(a):
    return a.ctypes.data_as(_p)


def _sig(name, res, *args):
    f = getattr(lib, name)
    f.restype = res
    f.argtypes = list(args)
    return f


_sig("orc_low_pass", C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, _p)
_sig("orc_low_pass_2", C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, _p)
_sig("orc_complex_band_pass", C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, _p)
_sig("orc_root_raised_cosine", C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, _p)
_sig("orc_gaussian", C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, _p)
_sig("orc_frontend_taps", C.c_int, C.c_int, _p)
_sig("orc_set_ted_modmm", None, C.c_int, C.c_int)
_sig("orc_get_ted_modmm", None, _p, _p)
_sig("orc_mmse_table", _p)
_sig("orc_atan_table", _p)
_sig("orc_tanh_table", _p)
_sig("orc_sincosf", None, C.c_float, _p, _p)
_sig("orc_sincos_turn", None, C.c_uint64, _p, _p)
_sig("orc_fast_atan2f", C.c_float, C.c_float, C.c_float)
_sig("orc_phase_inc_to_turn", C.c_uint64, C.c_double)
_sig("orc_rotator", None, _p, _sz, C.c_uint64, C.c_uint64, _p)
_sig("orc_decim_count", _sz, _sz, C.c_int, C.c_int)
_sig("orc_decim_fir_ccf", _sz, _p, _sz, _p, C.c_int, C.c_int, C.c_int, _p)
_sig("orc_decim_fir_ccf_m16", _sz, _p, _sz, _p, C.c_int, C.c_int, _p)
_sig("orc_decim_auto", _sz, _p, _sz, _p, C.c_int, C.c_int, _p)
_sig("orc_decim_uses_m16", C.c_int, C.c_int, C.c_int)
_sig("orc_decim_fir_ccf_pl", _sz, _p, _sz, _p, C.c_int, C.c_int, _p)
_sig("orc_decim_uses_pl", C.c_int, C.c_int, C.c_int)
_sig("orc_decim_fir_ccf_pm", _sz, _p, _sz, _p, C.c_int, C.c_int, _p)
_sig("orc_decim_uses_pm", C.c_int, C.c_int, C.c_int)
_sig("orc_decim_fir_ccf_simd", _sz, _p, _sz, _p, C.c_int, C.c_int, _p)
_sig("orc_set_decim_impl", None, C.c_int)
_sig("orc_m16_steps", C.c_int, C.c_int, C.c_int)
_sig("orc_resamp_ccf", _sz, _p, _sz, _p, C.c_int, C.c_int, C.c_int, _p)
_sig("orc_fir_ccf", None, _p, _sz, _p, C.c_int, _p)
_sig("orc_fir_fff", None, _p, _sz, _p, C.c_int, _p)
_sig("orc_fir_ccc", None, _p, _sz, _p, C.c_int, _p)
_sig("orc_fir_ccc_conj_pair", None, _p, _sz, _p, _p, C.c_int, _p, _p)
_sig("orc_quad_demod", None, _p, _sz, C.c_float, _p)
_sig("orc_symbol_sync_ff", _sz, _p, _sz, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, _p)
_sig("orc_soft_quant", None, _p, _sz, C.c_float, C.c_float, _p)
_sig("orc_cc_decode_k7", _sz, _p, _sz, _p)
_sig("orc_cc_encode_k7", None, _p, _sz, _p)
_sig("orc_descramble", None, _p, _sz, C.c_uint32, C.c_uint32, C.c_int, _p)
_sig("orc_scramble", None, _p, _sz, C.c_uint32, C.c_uint32, C.c_int, _p)
_sig("orc_frontend", _sz, _p, _sz, C.c_int, C.c_double, _p)
_sig("orc_demod_2fsk", None, _p, _sz, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p)
_sig("orc_demod_gmsk", None, _p, _sz, C.c_int, C.c_int, C.c_int, C.c_int, _p)
_sig("orc_demod_qpsk", None, _p, _sz, C.c_int, C.c_int, C.c_int, C.c_int, _p)
_sig("orc_demod_out_free", None, _p)
_sig("orc_mod_2fsk", _sz, _p, _sz, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p)
_sig("orc_mod_gmsk", _sz, _p, _sz, C.c_int, C.c_int, C.c_int, C.c_int, _p)
_sig("orc_mod_qpsk", _sz, _p, _sz, C.c_int, C.c_int, C.c_int, C.c_int, _p)
_sig("orc_tx_interp", _sz, _p, _sz, C.c_int, _p)
_sig("orc_demod_dmr", None, _p, _sz, C.c_int, C.c_int, _p)
_sig("orc_chan_proto_taps", C.c_int, C.c_int, _p)
_sig("orc_pfb_channelizer", _sz, _p, _sz, _p, C.c_int, C.c_int, _p)
_sig("orc_demod_mmdvm_multi", _sz, _p, _sz, C.c_int, _p, _sz)
_sig("orc_demod_4fsk", None, _p, _sz, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p)
_sig("orc_demod_bpsk", None, _p, _sz, C.c_int, C.c_int, C.c_int, C.c_int, _p)
_sig("orc_mod_4fsk", _sz, _p, _sz, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p)
_sig("orc_mod_bpsk", _sz, _p, _sz, C.c_int, C.c_int, C.c_int, C.c_int, _p)
_sig("orc_clock_recovery_mm_cc", _sz, _p, _sz, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _p)
_sig("orc_modem_sync", _sz, C.c_int, _p, _sz, _p, _p, _p)
_sig("orc_modem_sync_geometry", C.c_int, C.c_int, _p, _p)
_sig("orc_modem_sync_collected", _sz)
_sig("orc_deframer", _sz, C.c_int, _p, _sz, _p, _p)
_sig("orc_rssi_tag", _sz, _p, _sz, C.c_float, _p)
_sig("orc_demod_mmdvm", _sz, _p, _sz, C.c_int, C.c_int, _p, _sz, _p, C.c_float, _p)
_sig("orc_demod_mmdvm_multi_rssi", _sz, _p, _sz, C.c_int, _p, _sz, _p, _sz, C.c_float)
_sig("orc_demod_mmdvm_multi_4fsk", _sz, _p, _sz, C.c_int, _p, _sz, _p, _sz, C.c_float, _p, _sz, _p)
_sig("orc_demod_mmdvm_xlating", _sz, _p, _sz, C.c_int, C.c_int, C.c_int, C.c_int, _p, _sz, _p, _sz, C.c_float)
_sig("orc_mod_mmdvm", _sz, _p, _sz, C.c_int, C.c_float, _p)
_sig("orc_mod_mmdvm_multi", _sz, _p, _sz, C.c_int, C.c_int, _p)
_sig("orc_batch_rx", C.c_double, C.c_int, _p, C.c_int, _sz, C.c_int, C.c_double, C.c_int, _p)

WIN_HAMMING, WIN_HANN, WIN_BLACKMAN, WIN_RECT, WIN_BH = 0, 1, 2, 3, 5
MODE_2FSK_1K, MODE_GMSK_10K, MODE_QPSK_250K = 0, 1, 2


def low_pass(gain, fs, fc, tw, win=WIN_HAMMING):
    n = lib.orc_low_pass(gain, fs, fc, tw, win, None)
    t = np.zeros(n, np.float32)
    lib.orc_low_pass(gain, fs, fc, tw, win, _ptr(t))
    return t


def low_pass_2(gain, fs, fc, tw, att, win=WIN_HAMMING):
    n = lib.orc_low_pass_2(gain, fs, fc, tw, att, win, None)
    t = np.zeros(n, np.float32)
    lib.orc_low_pass_2(gain, fs, fc, tw, att, win, _ptr(t))
    return t


def complex_band_pass(gain, fs, lo, hi, tw, win=WIN_HAMMING):
    n = lib.orc_complex_band_pass(gain, fs, lo, hi, tw, win, None)
    t = np.zeros(n, cf32)
    lib.orc_complex_band_pass(gain, fs, lo, hi, tw, win, _ptr(t))
    return t


def complex_band_pass_2(gain, fs, lo, hi, tw, att, win=WIN_HAMMING):
    lib.orc_complex_band_pass_2.restype = C.c_int
    args = (C.c_double(gain), C.c_double(fs), C.c_double(lo), C.c_double(hi), C.c_double(tw), C.c_double(att), win)
    n = lib.orc_complex_band_pass_2(*args, None)
    t = np.zeros(n, cf32)
    lib.orc_complex_band_pass_2(*args, _ptr(t))
    return t


def root_raised_cosine(gain, fs, sr, alpha, ntaps):
    n = lib.orc_root_raised_cosine(gain, fs, sr, alpha, ntaps, None)
    t = np.zeros(n, np.float32)
    lib.orc_root_raised_cosine(gain, fs, sr, alpha, ntaps, _ptr(t))
    return t


def gaussian(gain, spb, bt, ntaps):
    t = np.zeros(ntaps, np.float32)
    lib.orc_gaussian(gain, spb, bt, ntaps, _ptr(t))
    return t


def frontend_taps(samp_rate):
    n = lib.orc_frontend_taps(samp_rate, None)
    t = np.zeros(n, np.float32)
    if n:
        lib.orc_frontend_taps(samp_rate, _ptr(t))
    return t


def table(name, n):
    p = getattr(lib, "orc_%s_table" % name)()
    return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(n,)).copy()


def sincosf(x):
    s, c = C.c_float(), C.c_float()
    lib.orc_sincosf(x, C.byref(s), C.byref(c))
    return s.value, c.value


def sincos_turn(a):
    s, c = C.c_float(), C.c_float()
    lib.orc_sincos_turn(a, C.byref(s), C.byref(c))
    return s.value, c.value


def phase_inc_to_turn(rad):
    return lib.orc_phase_inc_to_turn(rad)


def rotator(x, inc, acc0=0):
    x = np.ascontiguousarray(x, cf32)
    y = np.empty_like(x)
    lib.orc_rotator(_ptr(x), x.size, inc, acc0, _ptr(y))
    return y


def decim_fir_ccf(x, taps, decim, nsplit=4):
    x = np.ascontiguousarray(x, cf32)
    taps = np.ascontiguousarray(taps, np.float32)
    n = lib.orc_decim_count(x.size, 1, decim)
    y = np.empty(n, cf32)
    lib.orc_decim_fir_ccf(_ptr(x), x.size, _ptr(taps), taps.size, decim, nsplit, _ptr(y))
    return y


def decim_fir_ccf_m16(x, taps, decim):
    x = np.ascontiguousarray(x, cf32)
    taps = np.ascontiguousarray(taps, np.float32)
    n = lib.orc_decim_count(x.size, 1, decim)
    y = np.empty(n, cf32)
    lib.orc_decim_fir_ccf_m16(_ptr(x), x.size, _ptr(taps), taps.size, decim, _ptr(y))
    return y


def decim_fir_ccf_pl(x, taps, decim):
    x = np.ascontiguousarray(x, cf32)
    taps = np.ascontiguousarray(taps, np.float32)
    n = lib.orc_decim_count(x.size, 1, decim)
    y = np.empty(n, cf32)
    lib.orc_decim_fir_ccf_pl(_ptr(x), x.size, _ptr(taps), taps.size, decim, _ptr(y))
    return y


def decim_fir_ccf_pm(x, taps, decim):
    x = np.ascontiguousarray(x, cf32)
    taps = np.ascontiguousarray(taps, np.float32)
    y = np.empty(lib.orc_decim_count(x.size, 1, decim), cf32)
    lib.orc_decim_fir_ccf_pm(_ptr(x), x.size, _ptr(taps), taps.size, decim, _ptr(y))
    return y


def decim_fir_ccf_simd(x, taps, decim):
    x = np.ascontiguousarray(x, cf32)
    taps = np.ascontiguousarray(taps, np.float32)
    n = lib.orc_decim_count(x.size, 1, decim)
    y = np.empty(n, cf32)
    lib.orc_decim_fir_ccf_simd(_ptr(x), x.size, _ptr(taps), taps.size, decim, _ptr(y))
    return y


def decim_auto(x, taps, decim):
    x = np.ascontiguousarray(x, cf32)
    taps = np.ascontiguousarray(taps, np.float32)
    n = lib.orc_decim_count(x.size, 1, decim)
    y = np.empty(n, cf32)
    lib.orc_decim_auto(_ptr(x), x.size, _ptr(taps), taps.size, decim, _ptr(y))
    return y


def resamp_ccf(x, taps, interp, decim):
    x = np.ascontiguousarray(x, cf32)
    taps = np.ascontiguousarray(taps, np.float32)
    n = lib.orc_decim_count(x.size, interp, decim)
    y = np.empty(n, cf32)
    lib.orc_resamp_ccf(_ptr(x), x.size, _ptr(taps), taps.size, interp, decim, _ptr(y))
    return y


def frontend(x, samp_rate, carrier_offset_hz):
    x = np.ascontiguousarray(x, cf32)
    y = np.empty(x.size, cf32)
    n = lib.orc_frontend(_ptr(x), x.size, samp_rate, float(carrier_offset_hz), _ptr(y))
    return y[:n].copy()


def _take(o):
    def arr(p, n, dt):
        if not p or n == 0:
            return np.zeros(0, dt)
        return np.frombuffer(C.string_at(p, n * np.dtype(dt).itemsize), dtype=dt).copy()
    r = dict(filtered=arr(o.filtered, o.n_filtered, cf32), constellation=arr(o.constellation, o.n_const, cf32),
             bits_a=arr(o.bits_a, o.n_bits_a, np.uint8), bits_b=arr(o.bits_b, o.n_bits_b, np.uint8))
    lib.orc_demod_out_free(C.byref(o))
    return r


def demod_2fsk(x, sps=10, samp_rate=1000000, carrier_freq=1700, filter_width=2000, fm=False):
    x = np.ascontiguousarray(x, cf32)
    o = DemodOut()
    lib.orc_demod_2fsk(_ptr(x), x.size, sps, samp_rate, carrier_freq, filter_width, int(fm), C.byref(o))
    return _take(o)


def demod_gmsk(x, sps=1, samp_rate=1000000, carrier_freq=1700, filter_width=20000):
    x = np.ascontiguousarray(x, cf32)
    o = DemodOut()
    lib.orc_demod_gmsk(_ptr(x), x.size, sps, samp_rate, carrier_freq, filter_width, C.byref(o))
    return _take(o)


def demod_qpsk(x, sps=2, samp_rate=1000000, carrier_freq=1700, filter_width=160000):
    x = np.ascontiguousarray(x, cf32)
    o = DemodOut()
    lib.orc_demod_qpsk(_ptr(x), x.size, sps, samp_rate, carrier_freq, filter_width, C.byref(o))
    return _take(o)


def demod_dmr(x, sps=5, samp_rate=1000000):
    x = np.ascontiguousarray(x, cf32)
    o = DemodOut()
    lib.orc_demod_dmr(_ptr(x), x.size, sps, samp_rate, C.byref(o))
    return _take(o)


def demod_dsss(x, sps=25, samp_rate=1000000, filter_width=150):
    x = np.ascontiguousarray(x, cf32)
    o = DemodOut()
    lib.orc_demod_dsss(_ptr(x), x.size, sps, samp_rate, filter_width, C.byref(o))
    return _take(o)


def dsss_taps(sps=25):
    n = lib.orc_dsss_taps(sps, None)
    t = np.zeros(n, np.float32)
    lib.orc_dsss_taps(sps, _ptr(t))
    return t


def dsss_decoder(x, sps=25):
    x = np.ascontiguousarray(x, cf32)
    out = np.zeros(x.size // (13 * sps) + 4, cf32)
    lib.orc_dsss_decoder.restype = C.c_size_t
    n = lib.orc_dsss_decoder(_ptr(x), C.c_size_t(x.size), sps, _ptr(out))
    return out[:n].copy()


def demod_m17(x, samp_rate=1000000, filter_width=9000):
    x = np.ascontiguousarray(x, cf32)
    o = DemodOut()
    lib.orc_demod_m17(_ptr(x), x.size, samp_rate, filter_width, C.byref(o))
    return _take(o)


def _mod(fn, data, *args):
    data = np.ascontiguousarray(data, np.uint8)
    n = fn(_ptr(data), data.size, *args, None)
    y = np.zeros(n, cf32)
    m = fn(_ptr(data), data.size, *args, _ptr(y))
    return y[:m]


def mod_2fsk(data, sps=50, samp_rate=1000000, carrier_freq=1700, filter_width=2000, fm=False):
    return _mod(lib.orc_mod_2fsk, data, sps, samp_rate, carrier_freq, filter_width, int(fm))


def mod_gmsk(data, sps=10, samp_rate=1000000, carrier_freq=1700, filter_width=20000):
    return _mod(lib.orc_mod_gmsk, data, sps, samp_rate, carrier_freq, filter_width)


def mod_qpsk(data, sps=4, samp_rate=1000000, carrier_freq=1700, filter_width=160000):
    return _mod(lib.orc_mod_qpsk, data, sps, samp_rate, carrier_freq, filter_width)


def demod_4fsk(x, sps=5, samp_rate=1000000, carrier_freq=1700, filter_width=3000, fm=True):
    x = np.ascontiguousarray(x, cf32)
    o = DemodOut()
    lib.orc_demod_4fsk(_ptr(x), x.size, sps, samp_rate, carrier_freq, filter_width, int(fm), C.byref(o))
    return _take(o)


def demod_bpsk(x, sps=10, samp_rate=1000000, carrier_freq=1700, filter_width=1300):
    x = np.ascontiguousarray(x, cf32)
    o = DemodOut()
    lib.orc_demod_bpsk(_ptr(x), x.size, sps, samp_rate, carrier_freq, filter_width, C.byref(o))
    return _take(o)


def mod_4fsk(data, sps=25, samp_rate=1000000, carrier_freq=1700, filter_width=3500, fm=True):
    return _mod(lib.orc_mod_4fsk, data, sps, samp_rate, carrier_freq, filter_width, int(fm))


def mod_bpsk(data, sps=500, samp_rate=1000000, carrier_freq=1700, filter_width=1500):
    return _mod(lib.orc_mod_bpsk, data, sps, samp_rate, carrier_freq, filter_width)


class ModemSync:
    """gr_modem::synchronize/findSync/packBytes over successive calls; feed() returns the list of (frame_type, payload bytes)"""

    def __init__(self, modem_type):
        self.modem_type = modem_type
        self.st = np.zeros(5, np.uint32)
        self.bitbuf = np.zeros(3123 * 8 + 8, np.uint8)

    def feed_raw(self, bits):
        bits = np.ascontiguousarray(bits, np.uint8)
        out = np.zeros(2 * bits.size + 4096, np.uint8)
        n = lib.orc_modem_sync(self.modem_type, _ptr(bits), bits.size, _ptr(self.st), _ptr(self.bitbuf), _ptr(out))
        self.collected = int(lib.orc_modem_sync_collected())   # > 0 <=> gr_modem::synchronize's data_to_process for these bits
        return out[:n].copy()

    def feed(self, bits):
        return parse_frames(self.feed_raw(bits))


def parse_frames(raw, with_sync=False):
    """records { u32 frame_type, u32 nbytes | _modem_sync << 16, payload padded to 4 } -> [(frame_type, bytes)] (+ _modem_sync)"""
    frames, pos = [], 0
    raw = np.ascontiguousarray(raw, np.uint8)
    while pos + 8 <= raw.size:
        ft, w = np.frombuffer(raw[pos:pos + 8].tobytes(), np.uint32)
        nb, msync = int(w) & 0xFFFF, int(w) >> 16
        rec = (int(ft), raw[pos + 8:pos + 8 + nb].tobytes())
        frames.append(rec + (msync,) if with_sync else rec)
        pos += 8 + ((nb + 3) & ~3)
    return frames


def deframer(type_, bits, state=None):
    """gr_deframer_bb over one call; state = np.uint32[3] carried between calls (None: fresh)"""
    bits = np.ascontiguousarray(bits, np.uint8)
    st = np.zeros(3, np.uint32) if state is None else state
    out = np.zeros(2 * bits.size + 32, np.uint8)
    n = lib.orc_deframer(type_, _ptr(bits), bits.size, _ptr(st), _ptr(out))
    return out[:n].copy()


def rssi_tag(x, cal=0.0):
    x = np.ascontiguousarray(x, cf32)
    db = np.zeros(x.size // 300 + 1, np.float32)
    n = lib.orc_rssi_tag(_ptr(x), x.size, cal, _ptr(db))
    return db[:n].copy()


def demod_mmdvm(x, samp_rate=250000, filter_width=5000, cal=0.0):
    x = np.ascontiguousarray(x, cf32)
    cap = x.size * 12 // 125 + 4
    out = np.zeros(cap, np.int16)
    rssi = np.zeros(cap // 300 + 2, np.float32)
    nr = C.c_size_t(0)
    n = lib.orc_demod_mmdvm(_ptr(x), x.size, samp_rate, filter_width, _ptr(out), cap, _ptr(rssi), cal, C.byref(nr))
    return out[:n].copy(), rssi[:nr.value].copy()


def demod_mmdvm_multi_rssi(x, M, cal=0.0):
    x = np.ascontiguousarray(x, cf32)
    cap = (x.size // M) * 24 // 25 + 4
    out = np.zeros((M, cap), np.int16)
    rcap = cap // 300 + 2
    rssi = np.zeros((M, rcap), np.float32)
    n = lib.orc_demod_mmdvm_multi_rssi(_ptr(x), x.size, M, _ptr(out), cap, _ptr(rssi), rcap, cal)
    return out[:, :n].copy(), rssi[:, :n // 300].copy()


def demod_mmdvm_multi_4fsk(x, M):
    """channelizer + per-channel FM int16 + per-channel 4FSK dibits (list of uint8 arrays)"""
    x = np.ascontiguousarray(x, cf32)
    cap = (x.size // M) * 24 // 25 + 4
    out = np.zeros((M, cap), np.int16)
    dcap = 2 * (cap // 4 + 16)
    dib = np.zeros((M, dcap), np.uint8)
    nd = np.zeros(M, np.uint64)
    n = lib.orc_demod_mmdvm_multi_4fsk(_ptr(x), x.size, M, _ptr(out), cap, None, 0, 0.0, _ptr(dib), dcap, _ptr(nd))
    return out[:, :n].copy(), [dib[c, :int(nd[c])].copy() for c in range(M)]


def demod_mmdvm_multi_full(x, M, cal=0.0):
    """the whole C4 receiver as bench.py runs it: PFB channelizer + per-channel FM int16 + rssi_tag_block values + 4FSK dibits"""
    x = np.ascontiguousarray(x, cf32)
    cap = (x.size // M) * 24 // 25 + 4
    out = np.zeros((M, cap), np.int16)
    rcap = cap // 300 + 2
    rssi = np.zeros((M, rcap), np.float32)
    dcap = 2 * (cap // 4 + 16)
    dib = np.zeros((M, dcap), np.uint8)
    nd = np.zeros(M, np.uint64)
    n = lib.orc_demod_mmdvm_multi_4fsk(_ptr(x), x.size, M, _ptr(out), cap, _ptr(rssi), rcap, cal, _ptr(dib), dcap, _ptr(nd))
    return out[:, :n].copy(), rssi[:, :n // 300].copy(), [dib[c, :int(nd[c])].copy() for c in range(M)]


_sig("orc_demod_mmdvm_xlating_bank_4fsk", _sz, _p, _sz, C.c_int, _p, _sz, _p, _sz, C.c_float, _p, _sz, _p)
_sig("orc_mmdvm_channel_tails", _sz, _p, C.c_int, _sz, _p, _sz, _p, _sz, C.c_float, _p, _sz, _p)


def mmdvm_channel_tails(ch, cal=0.0):
    """per-channel chains of gr_demod_mmdvm_multi2 on channel streams ch [nch, n1] at 25 ksps: (int16 [nch, n2], rssi, dibit list)"""
    ch = np.ascontiguousarray(ch, cf32)
    nch, n1 = ch.shape
    cap = n1 * 24 // 25 + 4
    out = np.zeros((nch, cap), np.int16)
    rcap = cap // 300 + 2
    rssi = np.zeros((nch, rcap), np.float32)
    dcap = 2 * (cap // 4 + 16)
    dib = np.zeros((nch, dcap), np.uint8)
    nd = np.zeros(nch, np.uint64)
    n = lib.orc_mmdvm_channel_tails(_ptr(ch), nch, n1, _ptr(out), cap, _ptr(rssi), rcap, cal, _ptr(dib), dcap, _ptr(nd))
    return out[:, :n].copy(), rssi[:, :n // 300].copy(), [dib[c, :int(nd[c])].copy() for c in range(nch)]



def demod_mmdvm_xlating_bank_4fsk(x, N, cal=0.0):
    """BASELINE configs[3] literal: N freq-xlating FIR decimators 1:N + the multi2 per-channel chain + 4FSK dibits"""
    x = np.ascontiguousarray(x, cf32)
    cap = (x.size // N + 2) * 24 // 25 + 4
    out = np.zeros((N, cap), np.int16)
    rcap = cap // 300 + 2
    rssi = np.zeros((N, rcap), np.float32)
    dcap = 2 * (cap // 4 + 16)
    dib = np.zeros((N, dcap), np.uint8)
    nd = np.zeros(N, np.uint64)
    n = lib.orc_demod_mmdvm_xlating_bank_4fsk(_ptr(x), x.size, N, _ptr(out), cap, _ptr(rssi), rcap, cal, _ptr(dib), dcap, _ptr(nd))
    return out[:, :n].copy(), rssi[:, :n // 300].copy(), [dib[c, :int(nd[c])].copy() for c in range(N)]


def demod_mmdvm_xlating(x, N, separation=25000, D=10, fw=8000, cal=0.0):
    x = np.ascontiguousarray(x, cf32)
    cap = x.size // D + 4
    out = np.zeros((N, cap), np.int16)
    rcap = cap // 300 + 2
    rssi = np.zeros((N, rcap), np.float32)
    n = lib.orc_demod_mmdvm_xlating(_ptr(x), x.size, N, separation, D, fw, _ptr(out), cap, _ptr(rssi), rcap, cal)
    return out[:, :n].copy(), rssi[:, :n // 300].copy()


_sig("orc_set_zero_runs", None, _p, _sz)


def set_zero_runs(runs):
    """gr_zero_idle_bursts runs [(channel, start, count), ...] for the NEXT mod_mmdvm / mod_mmdvm_multi call (None = off)"""
    global _zr
    if not runs:
        lib.orc_set_zero_runs(None, 0)
        return
    _zr = np.ascontiguousarray(np.array(runs, np.uint64).reshape(-1, 3))
    lib.orc_set_zero_runs(_ptr(_zr), _zr.shape[0])


def mod_mmdvm(x, filter_width=5000, bb_gain=1.0):
    x = np.ascontiguousarray(x, np.int16)
    m = lib.orc_mod_mmdvm(_ptr(x), x.size, filter_width, bb_gain, None)
    y = np.zeros(m, cf32)
    lib.orc_mod_mmdvm(_ptr(x), x.size, filter_width, bb_gain, _ptr(y))
    return y


def mod_mmdvm_multi(x, filter_width=5000):
    """x: int16 [N][n] (24 ksps FM baseband per channel) -> 250 ksps cf32 of the 10-port synthesizer"""
    x = np.ascontiguousarray(x, np.int16)
    N, n = x.shape
    m = lib.orc_mod_mmdvm_multi(_ptr(x), n, N, filter_width, None)
    y = np.zeros(m, cf32)
    lib.orc_mod_mmdvm_multi(_ptr(x), n, N, filter_width, _ptr(y))
    return y


def tx_interp(x, samp_rate):
    x = np.ascontiguousarray(x, cf32)
    n = lib.orc_tx_interp(_ptr(x), x.size, samp_rate, None)
    y = np.zeros(n, cf32)
    m = lib.orc_tx_interp(_ptr(x), x.size, samp_rate, _ptr(y))
    return y[:m]


_sig("orc_fll_band_edge", None, _p, _sz, C.c_float, C.c_float, C.c_int, C.c_float, _p)
_sig("orc_costas", None, _p, _sz, C.c_float, C.c_int, C.c_int, _p)
_sig("orc_agc2", None, _p, _sz, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _p)
_sig("orc_symbol_sync_ff", _sz, _p, _sz, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, _p)
_sig("orc_symbol_sync_cc", _sz, _p, _sz, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, _p)
TED_MM, TED_MOD_MM = 0, 1
CONST_BPSK, CONST_DQPSK, CONST_4LEVEL = 0, 1, 2


def fll_band_edge(x, sps, rolloff, ntaps, bw):
    x = np.ascontiguousarray(x, cf32)
    y = np.empty(x.size, cf32)
    lib.orc_fll_band_edge(_ptr(x), x.size, sps, rolloff, ntaps, bw, _ptr(y))
    return y


def costas(x, bw, order, use_snr):
    x = np.ascontiguousarray(x, cf32)
    y = np.empty(x.size, cf32)
    lib.orc_costas(_ptr(x), x.size, bw, order, int(use_snr), _ptr(y))
    return y


def agc2(x, attack, decay, ref, gain, max_gain=65536.0):
    x = np.ascontiguousarray(x, cf32)
    y = np.empty(x.size, cf32)
    lib.orc_agc2(_ptr(x), x.size, attack, decay, ref, gain, max_gain, _ptr(y))
    return y


_sig("orc_loop_clamp_hits", _sz, C.c_int)


def loop_clamp_hits(reset=True):
    """limiter hits of the oracle's timing loops since the last reset (single-threaded test statistic)"""
    return int(lib.orc_loop_clamp_hits(1 if reset else 0))


def symbol_sync_ff(x, ted, sps, loop_bw, damping, ted_gain, max_dev, constellation):
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty(x.size, np.float32)
    n = lib.orc_symbol_sync_ff(_ptr(x), x.size, ted, sps, loop_bw, damping, ted_gain, max_dev, constellation, _ptr(y))
    return y[:n].copy()


def symbol_sync_cc(x, ted, sps, loop_bw, damping, ted_gain, max_dev, constellation):
    x = np.ascontiguousarray(x, cf32)
    y = np.empty(x.size, cf32)
    n = lib.orc_symbol_sync_cc(_ptr(x), x.size, ted, sps, loop_bw, damping, ted_gain, max_dev, constellation, _ptr(y))
    return y[:n].copy()


_sig("orc_dmo_init", None, _p)
_sig("orc_dmo_process", _sz, _p, _p, _p, _sz, _p, _sz)
_sig("orc_golay1987_table", None, _p)
_sig("orc_golay1987_syndrome", C.c_uint32, C.c_uint32)
DMO_STATE_BYTES = 4 * 5 + 2 * 5 + 2 + 4 * 9 + 8 + 4 * 1440   # sizeof(orc_dmo_state) with natural alignment (checked below)


def golay1987_table():
    t = np.zeros(2048, np.uint32)
    lib.orc_golay1987_table(_ptr(t))
    return t


_sig("orc_demod_dmr_port3", _sz, _p, _sz, C.c_int, _p)


def demod_dmr_port3(x, samp_rate=1000000):
    x = np.ascontiguousarray(x, cf32)
    n = lib.orc_demod_dmr_port3(_ptr(x), x.size, samp_rate, None)
    y = np.empty(n, np.float32)
    lib.orc_demod_dmr_port3(_ptr(x), x.size, samp_rate, _ptr(y))
    return y


class DmoSink:
    """gr_dmr_dmo_sink restated (oracle/orc_dmr.c): process(float samples at 24 ksps) -> list of (type, fn, colour code, 33 bytes)"""

    def __init__(self):
        self.state = np.zeros(DMO_STATE_BYTES + 64, np.uint8)
        lib.orc_dmo_init(_ptr(self.state))
        self.table = golay1987_table()

    def process(self, x, cap=64):
        x = np.ascontiguousarray(x, np.float32)
        out = np.zeros(40 * cap, np.uint8)
        n = lib.orc_dmo_process(_ptr(self.state), _ptr(self.table), _ptr(x), x.size, _ptr(out), cap)
        assert n <= cap
        return [(int(out[40 * i]), int(out[40 * i + 1]), int(out[40 * i + 2]), out[40 * i + 4:40 * i + 37].tobytes()) for i in range(n)]


def cc_decode_k7(soft):
    soft = np.ascontiguousarray(soft, np.uint8)
    out = np.zeros(soft.size // 2 + 80, np.uint8)
    n = lib.orc_cc_decode_k7(_ptr(soft), soft.size, _ptr(out))
    return out[:n].copy()


def cc_encode_k7(bits):
    bits = np.ascontiguousarray(bits, np.uint8)
    out = np.zeros(2 * bits.size, np.uint8)
    lib.orc_cc_encode_k7(_ptr(bits), bits.size, _ptr(out))
    return out


def scramble(bits, mask=0x8A, seed=0x7F, length=7):
    bits = np.ascontiguousarray(bits, np.uint8)
    out = np.zeros_like(bits)
    lib.orc_scramble(_ptr(bits), bits.size, mask, seed, length, _ptr(out))
    return out


def descramble(bits, mask=0x8A, seed=0x7F, length=7):
    bits = np.ascontiguousarray(bits, np.uint8)
    out = np.zeros_like(bits)
    lib.orc_descramble(_ptr(bits), bits.size, mask, seed, length, _ptr(out))
    return out


def chan_proto_taps(M):
    n = lib.orc_chan_proto_taps(M, None)
    t = np.zeros(n, np.float32)
    lib.orc_chan_proto_taps(M, _ptr(t))
    return t


def pfb_channelizer(x, taps, M):
    x = np.ascontiguousarray(x, cf32)
    taps = np.ascontiguousarray(taps, np.float32)
    out = np.zeros((M, x.size // M), cf32)
    lib.orc_pfb_channelizer(_ptr(x), x.size, _ptr(taps), taps.size, M, _ptr(out))
    return out


def demod_mmdvm_multi(x, M):
    x = np.ascontiguousarray(x, cf32)
    cap = (x.size // M) * 24 // 25 + 4
    out = np.zeros((M, cap), np.int16)
    n = lib.orc_demod_mmdvm_multi(_ptr(x), x.size, M, _ptr(out), cap)
    return out[:, :n].copy()


def fir_ccc(x, taps):
    x = np.ascontiguousarray(x, cf32); taps = np.ascontiguousarray(taps, cf32)
    out = np.zeros(x.size, cf32)
    lib.orc_fir_ccc(_ptr(x), x.size, _ptr(taps), taps.size, _ptr(out))
    return out


def fir_ccc_conj_pair(x, up, lo):
    """-> (upper, lower) outputs of the 2FSK discriminator's filter pair (shared real-tap chains when lo == conj(up) bit for bit)"""
    x = np.ascontiguousarray(x, cf32); up = np.ascontiguousarray(up, cf32); lo = np.ascontiguousarray(lo, cf32)
    ou, ol = np.zeros(x.size, cf32), np.zeros(x.size, cf32)
    lib.orc_fir_ccc_conj_pair(_ptr(x), x.size, _ptr(up), _ptr(lo), up.size, _ptr(ou), _ptr(ol))
    return ou, ol


def batch_rx(mode, iq, samp_rate, carrier_offset_hz=0.0, threads=0):
    iq = np.ascontiguousarray(iq, cf32)
    batch, n = iq.shape
    chk = C.c_uint64()
    t = lib.orc_batch_rx(mode, _ptr(iq), batch, n, samp_rate, float(carrier_offset_hz), threads, C.byref(chk))
    return t, chk.value


_sig("orc_pipeline_rx_2fsk1k", C.c_double, _p, C.c_int, C.c_size_t, C.c_double, C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_int))


def pipeline_rx_2fsk1k(iq, carrier_offset_hz=0.0):
    """SURVEY 8(d) CPU baseline (ii) emulated: [batch, n] streams at 1 Msps through a thread-per-block pipeline of the 2FSK-1k receiver
    (oracle/orc_pipeline.c).  Returns (wall seconds, checksum as batch_rx gives it, [seconds every stage worked])."""
    iq = np.ascontiguousarray(iq, cf32)
    batch, n = iq.shape
    chk, ns = C.c_uint64(), C.c_int()
    busy = (C.c_double * 16)()
    t = lib.orc_pipeline_rx_2fsk1k(_ptr(iq), batch, n, float(carrier_offset_hz), C.byref(chk), busy, C.byref(ns))
    return t, chk.value, [busy[k] for k in range(ns.value)]


_sig("orc_block_timing_enable", None, C.c_int)
_sig("orc_block_timing_count", C.c_int)
_sig("orc_block_timing_get", C.c_double, C.c_int, C.c_char_p, C.c_size_t)


def block_times(mode, x, samp_rate, carrier_offset_hz=0.0):
    """Run time of every primitive call of one single-thread chain run over one stream: [(block name, seconds)] in call order."""
    x = np.ascontiguousarray(x, cf32).reshape(1, -1)
    lib.orc_block_timing_enable(1)
    try:
        batch_rx(mode, x, samp_rate, carrier_offset_hz, 1)
    finally:
        lib.orc_block_timing_enable(0)
    out = []
    buf = C.create_string_buffer(64)
    for i in range(lib.orc_block_timing_count()):
        secs = lib.orc_block_timing_get(i, buf, 64)
        out.append((buf.value.decode(), float(secs)))
    return out


# ---- side outputs of gr_demod_base (oracle/orc_side.c)
_sig("orc_det_log2f", C.c_float, C.c_float)
_sig("orc_rssi_block", None, _p, C.c_size_t, C.c_float, _p)
_sig("orc_power_spectrum", None, _p, _p, C.c_size_t, _p)


def det_log2f(x):
    return float(lib.orc_det_log2f(float(x)))


ANALOG_KINDS = {"nbfm": 0, "am": 1, "wbfm": 2}


def ctcss_squelch_ff(x, rate=8000, freq=88.5, level=0.01, length=8000, ramp=160, gate=True):
    x = np.ascontiguousarray(x, np.float32)
    out = np.zeros(max(x.size, 1), np.float32)
    lib.orc_ctcss_squelch_ff.restype = C.c_size_t
    n = lib.orc_ctcss_squelch_ff(_ptr(x), C.c_size_t(x.size), rate, C.c_float(freq), C.c_double(level), length, ramp, int(gate), _ptr(out))
    return out[:n].copy()


def demod_analog(x, kind, samp_rate=1000000, filter_width=5000, ctcss=0.0, set_width=0):
    """-> dict(filtered=cf32 port 0, audio=f32 port 1); ctcss != 0 (NBFM): gr_demod_nbfm::set_ctcss(tone) was called;
    set_width != 0: gr_demod_X::set_filter_width(set_width) was called on the block constructed with filter_width"""
    x = np.ascontiguousarray(x, cf32)
    lib.orc_set_ctcss(C.c_float(ctcss))
    lib.orc_set_rx_filter_width(int(set_width))
    f, a = C.c_void_p(), C.c_void_p()
    nf, na = C.c_size_t(), C.c_size_t()
    lib.orc_demod_analog(_ptr(x), C.c_size_t(x.size), ANALOG_KINDS[kind], samp_rate, filter_width,
                         C.byref(f), C.byref(nf), C.byref(a), C.byref(na))
    filt = np.ctypeslib.as_array(C.cast(f, C.POINTER(C.c_float)), (2 * nf.value,)).copy().view(cf32) if nf.value else np.zeros(0, cf32)
    aud = np.ctypeslib.as_array(C.cast(a, C.POINTER(C.c_float)), (na.value,)).copy() if na.value else np.zeros(0, np.float32)
    lib.orc_free(f); lib.orc_free(a)
    lib.orc_set_ctcss(C.c_float(0.0))
    lib.orc_set_rx_filter_width(0)
    return dict(filtered=filt, audio=aud)


def mod_nbfm(audio, sps=20, samp_rate=1000000, filter_width=5000, bb_gain=1.0, ctcss=0.0, set_width=0):
    """ctcss > 0: gr_mod_nbfm::set_ctcss(tone) was called; < 0: set_ctcss(0) after it had been on (x0.98); 0: the constructor's graph;
    set_width != 0: gr_mod_nbfm::set_filter_width(set_width) was called"""
    audio = np.ascontiguousarray(audio, np.float32)
    lib.orc_mod_nbfm.restype = C.c_size_t
    args = (_ptr(audio), C.c_size_t(audio.size), sps, samp_rate, filter_width, C.c_float(bb_gain))
    lib.orc_set_tx_ctcss(C.c_float(ctcss))
    lib.orc_set_tx_filter_width(int(set_width))
    try:
        n = lib.orc_mod_nbfm(*args, None)
        y = np.zeros(n, cf32)
        m = lib.orc_mod_nbfm(*args, _ptr(y))
    finally:
        lib.orc_set_tx_ctcss(C.c_float(0.0))
        lib.orc_set_tx_filter_width(0)
    return y[:m]


def sig_source_sin(fs, freq, ampl, n, k0=0, offset=0.0):
    out = np.zeros(n, np.float32)
    lib.orc_sig_source_sin(C.c_double(fs), C.c_double(freq), C.c_double(ampl), C.c_float(offset), C.c_uint64(k0), C.c_size_t(n), _ptr(out))
    return out


def sig_source_cos(fs, freq, ampl, n, k0=0):
    out = np.zeros(n, np.float32)
    lib.orc_sig_source_cos(C.c_double(fs), C.c_double(freq), C.c_double(ampl), C.c_uint64(k0), C.c_size_t(n), _ptr(out))
    return out


def mod_dsss(data, sps=25, samp_rate=1000000, filter_width=150, bb_gain=1.0):
    data = np.ascontiguousarray(data, np.uint8)
    lib.orc_mod_dsss.restype = C.c_size_t
    args = (_ptr(data), C.c_size_t(data.size), sps, samp_rate, filter_width, C.c_float(bb_gain))
    n = lib.orc_mod_dsss(*args, None)
    y = np.zeros(max(n, 1), cf32)
    m = lib.orc_mod_dsss(*args, _ptr(y))
    return y[:m]


def mod_m17(data, sps=125, samp_rate=1000000, filter_width=9000, bb_gain=1.0):
    data = np.ascontiguousarray(data, np.uint8)
    lib.orc_mod_m17.restype = C.c_size_t
    args = (_ptr(data), C.c_size_t(data.size), sps, samp_rate, filter_width, C.c_float(bb_gain))
    n = lib.orc_mod_m17(*args, None)
    y = np.zeros(max(n, 1), cf32)
    m = lib.orc_mod_m17(*args, _ptr(y))
    return y[:m]


def mod_dmr(data, sps=125, samp_rate=1000000, filter_width=5000, bb_gain=1.0, zero_runs=None):
    """gr_mod_dmr; zero_runs = [(T, count), ...] tags at the zero-idle block's 24 ksps input"""
    data = np.ascontiguousarray(data, np.uint8)
    lib.orc_mod_dmr.restype = C.c_size_t
    zr = np.zeros((0, 3), np.uint64) if not zero_runs else np.ascontiguousarray([[0, t, c] for t, c in zero_runs], np.uint64)
    args = (_ptr(data), C.c_size_t(data.size), sps, samp_rate, filter_width, C.c_float(bb_gain), _ptr(zr) if zr.size else None, C.c_size_t(zr.shape[0]))
    n = lib.orc_mod_dmr(*args, None)
    y = np.zeros(max(n, 1), cf32)
    m = lib.orc_mod_dmr(*args, _ptr(y))
    return y[:m]


def mod_ssb(audio, sb=0, sps=125, samp_rate=1000000, filter_width=2700, bb_gain=1.0, set_width=0):
    audio = np.ascontiguousarray(audio, np.float32)
    lib.orc_mod_ssb.restype = C.c_size_t
    args = (_ptr(audio), C.c_size_t(audio.size), sps, samp_rate, filter_width, sb, C.c_float(bb_gain))
    lib.orc_set_tx_filter_width(int(set_width))
    try:
        n = lib.orc_mod_ssb(*args, None)
        y = np.zeros(max(n, 1), cf32)
        m = lib.orc_mod_ssb(*args, _ptr(y)) if n else 0
    finally:
        lib.orc_set_tx_filter_width(0)
    return y[:m]


def mod_am(audio, sps=125, samp_rate=1000000, filter_width=5000, bb_gain=1.0, set_width=0):
    audio = np.ascontiguousarray(audio, np.float32)
    lib.orc_mod_am.restype = C.c_size_t
    args = (_ptr(audio), C.c_size_t(audio.size), sps, samp_rate, filter_width, C.c_float(bb_gain))
    lib.orc_set_tx_filter_width(int(set_width))
    try:
        n = lib.orc_mod_am(*args, None)
        y = np.zeros(max(n, 1), cf32)
        m = lib.orc_mod_am(*args, _ptr(y)) if n else 0
    finally:
        lib.orc_set_tx_filter_width(0)
    return y[:m]


def preemph_taps(sample_rate, tau=50e-6):
    a, b = (C.c_double * 2)(), (C.c_double * 2)()
    lib.orc_preemph_taps(sample_rate, C.c_double(tau), a, b)
    return list(a), list(b)


def demod_ssb(x, sb=0, samp_rate=1000000, filter_width=2700, set_width=0, gain=None):
    """set_width != 0: gr_demod_ssb::set_filter_width(set_width) was called; gain: gr_demod_ssb::set_gain(gain) (None = the constructor's 0.9)"""
    x = np.ascontiguousarray(x, cf32)
    f, a = C.c_void_p(), C.c_void_p()
    nf, na = C.c_size_t(), C.c_size_t()
    lib.orc_set_rx_filter_width(int(set_width))
    lib.orc_set_rx_gain(C.c_float(-1.0 if gain is None else gain))
    try:
        lib.orc_demod_ssb(_ptr(x), C.c_size_t(x.size), samp_rate, filter_width, sb, C.byref(f), C.byref(nf), C.byref(a), C.byref(na))
    finally:
        lib.orc_set_rx_filter_width(0)
        lib.orc_set_rx_gain(C.c_float(-1.0))
    filt = np.ctypeslib.as_array(C.cast(f, C.POINTER(C.c_float)), (2 * nf.value,)).copy().view(cf32) if nf.value else np.zeros(0, cf32)
    aud = np.ctypeslib.as_array(C.cast(a, C.POINTER(C.c_float)), (na.value,)).copy() if na.value else np.zeros(0, np.float32)
    lib.orc_free(f); lib.orc_free(a)
    return dict(filtered=filt, audio=aud)


def band_pass_2(gain, fs, lo, hi, tw, att, win=WIN_HAMMING):
    lib.orc_band_pass_2.restype = C.c_int
    args = (C.c_double(gain), C.c_double(fs), C.c_double(lo), C.c_double(hi), C.c_double(tw), C.c_double(att), win)
    n = lib.orc_band_pass_2(*args, None)
    t = np.zeros(n, np.float32)
    lib.orc_band_pass_2(*args, _ptr(t))
    return t


def pwr_squelch_cc(x, db=-140.0, alpha=0.01, ramp=0, gate=True):
    x = np.ascontiguousarray(x, cf32)
    out = np.zeros(max(x.size, 1), cf32)
    lib.orc_pwr_squelch_cc.restype = C.c_size_t
    n = lib.orc_pwr_squelch_cc(_ptr(x), C.c_size_t(x.size), C.c_double(db), C.c_double(alpha), ramp, int(gate), _ptr(out))
    return out[:n].copy()


def deemph_taps(sample_rate, tau=50e-6):
    a, b = (C.c_double * 2)(), (C.c_double * 2)()
    lib.orc_deemph_taps(sample_rate, C.c_double(tau), a, b)
    return list(a), list(b)


def rssi_block(x, level=0.0):
    x = np.ascontiguousarray(x, cf32)
    out = np.zeros(max(x.size, 1), np.float32)
    lib.orc_rssi_block(_ptr(x), x.size, level, _ptr(out))
    return out[:x.size].copy()


def power_spectrum(x, window):
    x = np.ascontiguousarray(x, cf32)
    w = np.ascontiguousarray(window, np.float32)
    assert x.size == w.size and x.size & (x.size - 1) == 0
    out = np.zeros(x.size, np.float32)
    lib.orc_power_spectrum(_ptr(x), _ptr(w), x.size, _ptr(out))
    return out
