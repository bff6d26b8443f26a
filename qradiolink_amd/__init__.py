"""qradiolink_amd — MI355X-native drop-in for QRadioLink's gr_modem RX DSP hot path.

This package is a thin ctypes binding over the C ABI of ``libqrl_hip.so`` (include/qrl_hip.h).
The compute path is hand-written HIP for gfx950; torch is used only to own device memory and
streams.  There is NO CPU fallback: importing works without a GPU (host-side filter design is
usable), but creating a demodulator without the HIP library or a device raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QRL_LIB_PATH") or os.path.join(_HERE, "libqrl_hip.so")   # QRL_LIB_PATH: timing-experiment builds only

# gr_modem_types (reference src/modem_types.h:5-50)
MODEM_2FSK2KFM, MODEM_2FSK1KFM, MODEM_2FSK2K, MODEM_2FSK1K, MODEM_2FSK10KFM = 15, 16, 17, 18, 19
MODEM_GMSK2K, MODEM_GMSK1K, MODEM_GMSK10K = 20, 21, 22
MODEM_QPSK250K = 26
MODEM_QPSK20K, MODEM_QPSKVIDEO, MODEM_QPSK2K = 1, 2, 7
MODEM_BPSK2K, MODEM_BPSK1K = 0, 24
MODEM_4FSK2K, MODEM_4FSK10KFM, MODEM_4FSK2KFM, MODEM_4FSK1KFM, MODEM_4FSK100K = 3, 4, 5, 6, 27
MODEM_BPSK8 = 25
MODEM_NBFM2500, MODEM_NBFM5000, MODEM_WBFM, MODEM_AM5000 = 8, 9, 10, 14
MODEM_USB2500, MODEM_LSB2500, MODEM_CW600USB = 11, 12, 13
MODEM_M17 = 40
MODEM_DMR = 41
OPT_OVERLAP, OPT_UNFUSED_DEC2, OPT_FLL_SLIM, OPT_GROUPED, OPT_INPUT_RESIDENT = 1, 2, 3, 4, 5
CHAN_OPT_LEGACY_PFB, CHAN_OPT_LEGACY_TAIL, CHAN_OPT_SERIAL_TAIL = 1, 2, 3
WIN_HAMMING, WIN_HANN, WIN_BLACKMAN, WIN_RECTANGULAR, WIN_BLACKMAN_HARRIS = 0, 1, 2, 3, 5


class QrlError(RuntimeError):
    pass


class _Config(C.Structure):
    _fields_ = [("modem_type", C.c_int), ("use_mode_defaults", C.c_int), ("sps", C.c_int), ("samp_rate", C.c_int),
                ("carrier_freq", C.c_int), ("filter_width", C.c_int), ("fm", C.c_int), ("device_samp_rate", C.c_int),
                ("carrier_offset_hz", C.c_double), ("batch", C.c_int), ("max_chunk", C.c_size_t),
                ("hip_stream", C.c_void_p), ("enable_side_outputs", C.c_int), ("time_domain_samp_rate", C.c_int), ("time_domain_filter_width", C.c_double)]


class _ModConfig(C.Structure):
    _fields_ = [("modem_type", C.c_int), ("use_mode_defaults", C.c_int), ("sps", C.c_int), ("samp_rate", C.c_int),
                ("carrier_freq", C.c_int), ("filter_width", C.c_int), ("fm", C.c_int), ("batch", C.c_int),
                ("max_bytes", C.c_size_t), ("hip_stream", C.c_void_p), ("bb_gain", C.c_float),
                ("device_samp_rate", C.c_int), ("carrier_offset_hz", C.c_double)]


class _ChanConfig(C.Structure):
    _fields_ = [("num_channels", C.c_int), ("channel_first", C.c_int), ("channel_count", C.c_int), ("batch", C.c_int),
                ("max_chunk", C.c_size_t), ("hip_stream", C.c_void_p), ("form", C.c_int), ("channel_separation", C.c_int),
                ("decimation", C.c_int), ("filter_width", C.c_int)]


class _SynthConfig(C.Structure):
    _fields_ = [("num_channels", C.c_int), ("filter_width", C.c_int), ("batch", C.c_int), ("max_samples", C.c_size_t),
                ("hip_stream", C.c_void_p), ("bb_gain", C.c_float), ("single_carrier", C.c_int)]


class _ZeroRun(C.Structure):
    _fields_ = [("stream", C.c_int), ("channel", C.c_int), ("start", C.c_uint64), ("count", C.c_uint64)]


class _AModConfig(C.Structure):
    _fields_ = [("modem_type", C.c_int), ("batch", C.c_int), ("max_samples", C.c_size_t), ("hip_stream", C.c_void_p), ("bb_gain", C.c_float),
                ("device_samp_rate", C.c_int), ("carrier_offset_hz", C.c_double)]


class _Out(C.Structure):
    _fields_ = [("filtered", C.c_void_p), ("filtered_cap", C.c_size_t), ("constellation", C.c_void_p),
                ("constellation_cap", C.c_size_t), ("bits_a", C.c_void_p), ("bits_cap", C.c_size_t),
                ("bits_b", C.c_void_p), ("counts", C.c_void_p), ("audio", C.c_void_p), ("audio_cap", C.c_size_t)]


_lib = None


def load_library():
    """Load libqrl_hip.so (built in-tree by __graft_entry__.build()).  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise QrlError("libqrl_hip.so not found at %s: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(there is no CPU fallback)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, sz = C.c_void_p, C.c_size_t
    lib.qrl_version.restype = C.c_char_p
    lib.qrl_last_error.restype = C.c_char_p
    lib.qrl_strerror.restype = C.c_char_p
    lib.qrl_strerror.argtypes = [C.c_int]
    lib.qrl_init.argtypes = [C.c_int, C.POINTER(vp)]
    lib.qrl_shutdown.argtypes = [vp]
    lib.qrl_demod_create.argtypes = [vp, C.POINTER(_Config), C.POINTER(vp)]
    lib.qrl_demod_destroy.argtypes = [vp]
    lib.qrl_demod_reset.argtypes = [vp]
    lib.qrl_demod_set_carrier_offset.argtypes = [vp, C.c_double]
    lib.qrl_demod_set_option.argtypes = [vp, C.c_int, C.c_int]
    lib.qrl_demod_set_dmo_output.argtypes = [vp, vp, sz, vp]
    lib.qrl_demod_out_caps.argtypes = [vp, sz, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]
    lib.qrl_demod_audio_cap.argtypes = [vp, sz, C.POINTER(sz)]
    lib.qrl_mod_samples_per_block.argtypes = [vp, C.POINTER(sz)]
    lib.qrl_mod_samples_per_block.restype = sz
    lib.qrl_amod_create.argtypes = [vp, vp, C.POINTER(vp)]
    lib.qrl_amod_destroy.argtypes = [vp]
    lib.qrl_amod_destroy.restype = None
    lib.qrl_amod_reset.argtypes = [vp]
    lib.qrl_amod_set_bb_gain.argtypes = [vp, C.c_float]
    lib.qrl_amod_set_ctcss.argtypes = [vp, C.c_float]
    lib.qrl_amod_samples_per_sample.argtypes = [vp]
    lib.qrl_amod_samples_per_sample.restype = sz
    lib.qrl_amod_last_count.argtypes = [vp]
    lib.qrl_amod_last_count.restype = sz
    lib.qrl_amod_out_cap.argtypes = [vp, sz]
    lib.qrl_amod_out_cap.restype = sz
    lib.qrl_amod_process.argtypes = [vp, vp, sz, sz, vp, sz]
    lib.qrl_amod_sync.argtypes = [vp]
    lib.qrl_amod_stream.argtypes = [vp]
    lib.qrl_amod_stream.restype = vp
    for name in ("qrl_bptc19696_decode", "qrl_bptc19696_encode", "qrl_m17_decode_frames", "qrl_m17_encode_frames"):
        getattr(lib, name).argtypes = [vp, vp, vp, sz, vp]
    lib.qrl_demod_set_squelch.argtypes = [vp, C.c_double]
    lib.qrl_demod_set_agc.argtypes = [vp, C.c_float, C.c_float]
    lib.qrl_demod_set_filter_width.argtypes = [vp, C.c_int]
    lib.qrl_demod_set_gain.argtypes = [vp, C.c_float]
    lib.qrl_amod_set_filter_width.argtypes = [vp, C.c_int]
    lib.qrl_amod_set_carrier_offset.argtypes = [vp, C.c_double]
    lib.qrl_amod_set_cw_k.argtypes = [vp, C.c_int]
    lib.qrl_demod_process.argtypes = [vp, vp, sz, sz, C.POINTER(_Out)]
    lib.qrl_demod_sync.argtypes = [vp]
    lib.qrl_rssi_create.argtypes = [vp, C.c_int, C.c_float, vp, C.POINTER(vp)]
    lib.qrl_rssi_destroy.argtypes = [vp]
    lib.qrl_rssi_reset.argtypes = [vp]
    lib.qrl_rssi_set_level.argtypes = [vp, C.c_float]
    lib.qrl_rssi_process.argtypes = [vp, vp, sz, sz, vp, sz, vp, sz, vp, vp]
    lib.qrl_rssi_sync.argtypes = [vp]
    lib.qrl_fft_create.argtypes = [vp, C.c_int, C.c_uint, C.c_int, vp, C.POINTER(vp)]
    lib.qrl_fft_destroy.argtypes = [vp]
    lib.qrl_fft_set_enabled.argtypes = [vp, C.c_int]
    lib.qrl_fft_set_fft_size.argtypes = [vp, C.c_uint]
    lib.qrl_fft_get_fft_size.argtypes = [vp]
    lib.qrl_fft_get_fft_size.restype = C.c_uint
    lib.qrl_fft_set_window_type.argtypes = [vp, C.c_int]
    lib.qrl_fft_get_window_type.argtypes = [vp]
    lib.qrl_fft_process.argtypes = [vp, vp, sz, sz]
    lib.qrl_fft_get_fft_data.argtypes = [vp, vp, sz, C.POINTER(C.c_uint)]
    lib.qrl_fft_sync.argtypes = [vp]
    lib.qrl_demod_stream_wait.argtypes = [vp, vp]
    lib.qrl_demod_set_ctcss.argtypes = [vp, C.c_float]
    lib.qrl_demod_time_domain_cap.argtypes = [vp, sz, C.POINTER(sz)]
    lib.qrl_demod_set_time_domain_output.argtypes = [vp, vp, sz, vp]
    lib.qrl_demod_stream.restype = vp
    lib.qrl_demod_stream.argtypes = [vp]
    lib.qrl_demod_internal_streams.argtypes = [vp, C.POINTER(vp)]
    lib.qrl_chan_internal_streams.argtypes = [vp, C.POINTER(vp)]
    lib.qrl_demod_profile.argtypes = [vp, C.c_int]
    lib.qrl_demod_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_char_p)]
    lib.qrl_demod_process_host.argtypes = [vp, vp, sz, sz, vp, vp, sz, vp]
    lib.qrl_mod_create.argtypes = [vp, C.POINTER(_ModConfig), C.POINTER(vp)]
    lib.qrl_mod_destroy.argtypes = [vp]
    lib.qrl_mod_reset.argtypes = [vp]
    lib.qrl_mod_set_bb_gain.argtypes = [vp, C.c_float]
    lib.qrl_mod_set_carrier_offset.argtypes = [vp, C.c_double]
    lib.qrl_mod_samples_per_byte.restype = sz
    lib.qrl_mod_samples_per_byte.argtypes = [vp]
    lib.qrl_mod_process.argtypes = [vp, vp, sz, sz, vp, sz]
    lib.qrl_mod_sync.argtypes = [vp]
    lib.qrl_mod_stream.restype = vp
    lib.qrl_mod_stream.argtypes = [vp]
    lib.qrl_chan_create.argtypes = [vp, C.POINTER(_ChanConfig), C.POINTER(vp)]
    lib.qrl_chan_destroy.argtypes = [vp]
    lib.qrl_chan_reset.argtypes = [vp]
    lib.qrl_chan_set_level.argtypes = [vp, C.c_float]
    lib.qrl_chan_calibrate_rssi.argtypes = [vp, C.c_float]
    lib.qrl_chan_set_rssi_output.argtypes = [vp, vp, sz, vp]
    lib.qrl_chan_set_4fsk_output.argtypes = [vp, vp, sz, vp, sz, vp]
    lib.qrl_chan_out_cap.restype = sz
    lib.qrl_chan_out_cap.argtypes = [vp, sz]
    lib.qrl_chan_process.argtypes = [vp, vp, sz, sz, vp, sz, vp]
    lib.qrl_chan_sync.argtypes = [vp]
    lib.qrl_chan_stream_wait.argtypes = [vp, vp]
    lib.qrl_chan_stream.argtypes = [vp]
    lib.qrl_chan_stream.restype = vp
    lib.qrl_chan_set_option.argtypes = [vp, C.c_int, C.c_int]
    lib.qrl_chan_channelize.argtypes = [vp, vp, sz, sz, vp, sz, C.c_int]
    lib.qrl_chan_process_channels.argtypes = [vp, vp, sz, sz, vp, sz, vp]
    lib.qrl_chan_wait_for.argtypes = [vp, vp]
    lib.qrl_chan_profile.argtypes = [vp, C.c_int]
    lib.qrl_chan_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_char_p)]
    lib.qrl_chan_profile_read_kernels.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    lib.qrl_synth_create.argtypes = [vp, C.POINTER(_SynthConfig), C.POINTER(vp)]
    lib.qrl_synth_destroy.argtypes = [vp]
    lib.qrl_synth_reset.argtypes = [vp]
    lib.qrl_synth_set_bb_gain.argtypes = [vp, C.c_float]
    lib.qrl_synth_add_zero_runs.argtypes = [vp, vp, sz]
    lib.qrl_mod_add_zero_runs.argtypes = [vp, vp, sz]
    lib.qrl_synth_out_cap.restype = sz
    lib.qrl_synth_out_cap.argtypes = [vp, sz]
    lib.qrl_synth_process.argtypes = [vp, vp, sz, sz, vp, sz, C.POINTER(sz)]
    lib.qrl_synth_sync.argtypes = [vp]
    lib.qrl_deframer_create.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(vp)]
    lib.qrl_deframer_destroy.argtypes = [vp]
    lib.qrl_deframer_reset.argtypes = [vp]
    lib.qrl_deframer_process.argtypes = [vp, vp, sz, sz, vp, sz, vp, sz, vp]
    lib.qrl_deframer_sync.argtypes = [vp]
    lib.qrl_framesync_create.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(vp)]
    lib.qrl_framesync_destroy.argtypes = [vp]
    lib.qrl_framesync_reset.argtypes = [vp]
    lib.qrl_framesync_frame_bytes.argtypes = [vp]
    lib.qrl_framesync_process.argtypes = [vp, vp, sz, sz, vp, sz, vp, sz, vp]
    lib.qrl_framesync_sync.argtypes = [vp]
    lib.qrl_framesync_set_activity_output.argtypes = [vp, vp]
    lib.qrl_firdes_low_pass.argtypes = [C.c_double] * 4 + [C.c_int, vp]
    lib.qrl_firdes_low_pass_2.argtypes = [C.c_double] * 5 + [C.c_int, vp]
    lib.qrl_firdes_complex_band_pass.argtypes = [C.c_double] * 5 + [C.c_int, vp]
    lib.qrl_firdes_root_raised_cosine.argtypes = [C.c_double] * 4 + [C.c_int, vp]
    for n in ("mmse", "atan", "tanh"):
        getattr(lib, "qrl_table_" + n).argtypes = [vp]
    lib.qrl_phase_inc_to_turn.restype = C.c_uint64
    lib.qrl_phase_inc_to_turn.argtypes = [C.c_double]
    _lib = lib
    return lib


EXPORTED_SYMBOLS = [
    "qrl_init", "qrl_shutdown", "qrl_strerror", "qrl_last_error", "qrl_version", "qrl_demod_create",
    "qrl_demod_destroy", "qrl_demod_reset", "qrl_demod_set_carrier_offset", "qrl_demod_set_option", "qrl_demod_set_dmo_output", "qrl_demod_stream_wait", "qrl_demod_out_caps",
    "qrl_demod_audio_cap", "qrl_demod_set_squelch", "qrl_demod_set_agc", "qrl_demod_set_filter_width", "qrl_demod_set_gain", "qrl_demod_set_ctcss", "qrl_demod_time_domain_cap", "qrl_demod_set_time_domain_output",
    "qrl_bptc19696_decode", "qrl_bptc19696_encode", "qrl_m17_decode_frames", "qrl_m17_encode_frames",
    "qrl_amod_create", "qrl_amod_destroy", "qrl_amod_reset", "qrl_amod_set_bb_gain", "qrl_amod_set_ctcss", "qrl_amod_set_filter_width", "qrl_amod_set_carrier_offset", "qrl_amod_set_cw_k", "qrl_amod_samples_per_sample", "qrl_amod_last_count", "qrl_amod_out_cap", "qrl_amod_process", "qrl_amod_sync", "qrl_amod_stream",
    "qrl_demod_process", "qrl_demod_sync", "qrl_demod_stream", "qrl_demod_internal_streams", "qrl_chan_internal_streams", "qrl_demod_process_host", "qrl_demod_profile",
    "qrl_demod_profile_read", "qrl_mod_create", "qrl_mod_destroy", "qrl_mod_reset", "qrl_mod_set_bb_gain", "qrl_mod_set_carrier_offset",
    "qrl_mod_samples_per_byte", "qrl_mod_samples_per_block", "qrl_mod_add_zero_runs", "qrl_mod_process", "qrl_mod_sync", "qrl_mod_stream", "qrl_chan_set_option", "qrl_chan_channelize", "qrl_chan_process_channels", "qrl_chan_wait_for", "qrl_chan_stream_wait", "qrl_chan_stream", "qrl_chan_profile", "qrl_chan_profile_read", "qrl_chan_profile_read_kernels", "qrl_debug_decim_prof", "qrl_debug_decim_prof_enable", "qrl_chan_create",
    "qrl_chan_destroy", "qrl_chan_reset", "qrl_chan_set_level", "qrl_chan_calibrate_rssi", "qrl_chan_set_rssi_output", "qrl_chan_set_4fsk_output", "qrl_chan_out_cap", "qrl_chan_process", "qrl_chan_sync",
    "qrl_synth_create", "qrl_synth_destroy", "qrl_synth_reset", "qrl_synth_set_bb_gain", "qrl_synth_add_zero_runs", "qrl_synth_out_cap", "qrl_synth_process",
    "qrl_synth_sync",
    "qrl_rssi_create", "qrl_rssi_destroy", "qrl_rssi_reset", "qrl_rssi_set_level", "qrl_rssi_process", "qrl_rssi_sync", "qrl_rssi_stream",
    "qrl_fft_create", "qrl_fft_destroy", "qrl_fft_set_enabled", "qrl_fft_set_fft_size", "qrl_fft_get_fft_size", "qrl_fft_set_window_type",
    "qrl_fft_get_window_type", "qrl_fft_process", "qrl_fft_get_fft_data", "qrl_fft_sync", "qrl_fft_stream",
    "qrl_deframer_create", "qrl_deframer_destroy", "qrl_deframer_reset", "qrl_deframer_process", "qrl_deframer_sync",
    "qrl_framesync_create", "qrl_framesync_destroy", "qrl_framesync_reset", "qrl_framesync_frame_bytes", "qrl_framesync_process",
    "qrl_framesync_sync", "qrl_framesync_set_activity_output",
    "qrl_firdes_low_pass",
    "qrl_firdes_low_pass_2", "qrl_firdes_complex_band_pass", "qrl_firdes_root_raised_cosine", "qrl_table_mmse",
    "qrl_table_atan", "qrl_table_tanh", "qrl_phase_inc_to_turn",
]


def _check(rc, what):
    if rc != 0:
        lib = load_library()
        raise QrlError("%s failed: %s (%s)" % (what, lib.qrl_strerror(rc).decode(), lib.qrl_last_error().decode()))


# ---------------------------------------------------------------- host-side design (no GPU needed)
def _taps(fn, count_args, dtype=np.float32, mult=1):
    lib = load_library()
    n = fn(*count_args, None)
    t = np.zeros(n * mult, np.float32)
    fn(*count_args, t.ctypes.data_as(C.c_void_p))
    return t.view(dtype) if dtype != np.float32 else t


def low_pass(gain, fs, fc, tw, win=WIN_HAMMING):
    return _taps(load_library().qrl_firdes_low_pass, (gain, fs, fc, tw, win))


def low_pass_2(gain, fs, fc, tw, att, win=WIN_HAMMING):
    return _taps(load_library().qrl_firdes_low_pass_2, (gain, fs, fc, tw, att, win))


def complex_band_pass(gain, fs, lo, hi, tw, win=WIN_HAMMING):
    return _taps(load_library().qrl_firdes_complex_band_pass, (gain, fs, lo, hi, tw, win), np.complex64, 2)


def root_raised_cosine(gain, fs, sr, alpha, ntaps):
    return _taps(load_library().qrl_firdes_root_raised_cosine, (gain, fs, sr, alpha, ntaps))


def table(name):
    n = {"mmse": 129 * 8, "atan": 257, "tanh": 256}[name]
    t = np.zeros(n, np.float32)
    getattr(load_library(), "qrl_table_" + name)(t.ctypes.data_as(C.c_void_p))
    return t


# ---------------------------------------------------------------- device path
class Context:
    """qrl_init/qrl_shutdown wrapper (one per process and device)."""

    def __init__(self, device=0):
        self.lib = load_library()
        self.h = C.c_void_p()
        _check(self.lib.qrl_init(device, C.byref(self.h)), "qrl_init")
        self.device = device

    def close(self):
        if self.h:
            self.lib.qrl_shutdown(self.h)
            self.h = C.c_void_p()


class Demod:
    """Batch RX demodulator: mirrors make_gr_demod_2fsk / make_gr_demod_gmsk (+ gr_demod_base front end).

    process(iq) takes a torch cuda tensor complex64 [batch, n] (device-resident) and returns a dict of
    torch tensors: filtered [B, cap] complex64, constellation [B, cap] complex64, bits_a/bits_b [B, cap]
    uint8 and counts [B, 4] int32 (valid lengths per port).  Same ports as the reference hier blocks
    (gr_demod_2fsk.cpp:19-37)."""

    def __init__(self, ctx, modem_type, batch, max_chunk, device_samp_rate=1000000, carrier_offset_hz=0.0,
                 side_outputs=True, stream=None, time_domain_samp_rate=0, time_domain_filter_width=0.0, input_resident=True, **explicit):
        import torch
        self.torch = torch
        self.ctx, self.lib = ctx, ctx.lib
        cfg = _Config()
        cfg.modem_type = modem_type
        cfg.use_mode_defaults = 0 if explicit else 1
        if explicit:
            cfg.sps = explicit["sps"]
            cfg.samp_rate = explicit.get("samp_rate", 1000000)
            cfg.carrier_freq = explicit.get("carrier_freq", 1700)
            cfg.filter_width = explicit["filter_width"]
            cfg.fm = int(explicit.get("fm", 0))
        cfg.device_samp_rate = device_samp_rate
        cfg.carrier_offset_hz = carrier_offset_hz
        cfg.batch = batch
        cfg.max_chunk = max_chunk
        cfg.hip_stream = stream
        cfg.enable_side_outputs = int(side_outputs)
        cfg.time_domain_samp_rate = time_domain_samp_rate       # gr_demod_base::set_time_sink_samp_rate / set_time_domain_filter_width (0: the constructor's 1:10)
        cfg.time_domain_filter_width = time_domain_filter_width
        self.batch, self.max_chunk, self.side = batch, max_chunk, side_outputs
        self.h = C.c_void_p()
        _check(self.lib.qrl_demod_create(ctx.h, C.byref(cfg), C.byref(self.h)), "qrl_demod_create")
        f, c, b = C.c_size_t(), C.c_size_t(), C.c_size_t()
        _check(self.lib.qrl_demod_out_caps(self.h, max_chunk, C.byref(f), C.byref(c), C.byref(b)), "qrl_demod_out_caps")
        self.caps = (f.value, c.value, b.value)
        a = C.c_size_t()
        _check(self.lib.qrl_demod_audio_cap(self.h, max_chunk, C.byref(a)), "qrl_demod_audio_cap")
        self.audio_cap = a.value      # > 0 for the analogue voice receivers (port 1 = audio)
        # process / process_async wait for the producer of iq on the host before they call the library, so the IQ of a call IS complete in
        # device memory: QRL_OPT_INPUT_RESIDENT (the front end's helper kernels on their own stream) is on unless the caller says otherwise
        if input_resident and stream is None:
            _check(self.lib.qrl_demod_set_option(self.h, OPT_INPUT_RESIDENT, 1), "qrl_demod_set_option")
        self.new_outputs()

    def new_outputs(self):
        """Fresh output tensors for the following calls (calls that are in flight together need their own: the serial tail of call
        k runs beside the front end of call k + 1).  Returns them as a dict; the previous set stays valid for its calls."""
        torch = self.torch
        f, c, b = self.caps
        dev = "cuda:%d" % self.ctx.device
        self.filtered = torch.zeros((self.batch, f), dtype=torch.complex64, device=dev) if self.side else None
        self.constellation = torch.zeros((self.batch, c), dtype=torch.complex64, device=dev) if self.side else None
        self.bits_a = torch.zeros((self.batch, b), dtype=torch.uint8, device=dev)
        self.bits_b = torch.zeros((self.batch, b), dtype=torch.uint8, device=dev)
        self.counts = torch.zeros((self.batch, 4), dtype=torch.int32, device=dev)
        self._out = _Out()
        if self.side:
            self._out.filtered, self._out.filtered_cap = self.filtered.data_ptr(), f
            self._out.constellation, self._out.constellation_cap = self.constellation.data_ptr(), c
        self._out.bits_a, self._out.bits_b, self._out.bits_cap = self.bits_a.data_ptr(), self.bits_b.data_ptr(), b
        self._out.counts = self.counts.data_ptr()
        self.audio = None
        if self.audio_cap:
            self.audio = torch.zeros((self.batch, self.audio_cap), dtype=torch.float32, device=dev)
            self._out.audio, self._out.audio_cap = self.audio.data_ptr(), self.audio_cap
        torch.cuda.current_stream().synchronize()   # the zero fills ran on torch's stream
        return self._ports()

    def _ports(self):
        d = dict(filtered=self.filtered, constellation=self.constellation, bits_a=self.bits_a, bits_b=self.bits_b, counts=self.counts)
        if self.audio is not None:
            d["audio"] = self.audio
        return d

    def process_async(self, iq):
        """Queue one pass over iq ([batch, n] complex64 cuda tensor) on the handle's stream."""
        assert iq.is_cuda and iq.dtype == self.torch.complex64 and iq.dim() == 2 and iq.shape[0] == self.batch
        assert iq.stride(1) == 1
        self.torch.cuda.current_stream().synchronize()   # the handle runs on its own HIP streams: whatever produced iq must be done
        _check(self.lib.qrl_demod_process(self.h, iq.data_ptr(), iq.stride(0), iq.shape[1], C.byref(self._out)),
               "qrl_demod_process")

    def sync(self):
        _check(self.lib.qrl_demod_sync(self.h), "qrl_demod_sync")

    def process(self, iq):
        self.process_async(iq)
        self.sync()
        return self._ports()

    def set_squelch(self, db):
        _check(self.lib.qrl_demod_set_squelch(self.h, C.c_double(db)), "qrl_demod_set_squelch")

    def enable_time_domain(self):
        """gr_demod_base::enable_time_domain(true): self.scope complex64 [batch, cap] / self.scope_counts int32 [batch] receive the 100 ksps
        scope items of every following call (qrl_demod_set_time_domain_output)"""
        t = self.torch
        cap = C.c_size_t()
        _check(self.lib.qrl_demod_time_domain_cap(self.h, self.max_chunk, C.byref(cap)), "qrl_demod_time_domain_cap")
        dev = self.bits_a.device
        self.scope = t.zeros((self.batch, cap.value), dtype=t.complex64, device=dev)
        self.scope_counts = t.zeros((self.batch,), dtype=t.int32, device=dev)
        t.cuda.current_stream().synchronize()
        _check(self.lib.qrl_demod_set_time_domain_output(self.h, self.scope.data_ptr(), cap.value, self.scope_counts.data_ptr()), "qrl_demod_set_time_domain_output")

    def set_ctcss(self, tone_hz):
        """gr_demod_nbfm::set_ctcss: 0 = off, else the CTCSS tone that opens the audio path"""
        _check(self.lib.qrl_demod_set_ctcss(self.h, C.c_float(tone_hz)), "qrl_demod_set_ctcss")

    def set_agc(self, attack, decay):
        _check(self.lib.qrl_demod_set_agc(self.h, C.c_float(attack), C.c_float(decay)), "qrl_demod_set_agc")

    def set_filter_width(self, width):
        """gr_demod_base::set_filter_width for this handle's analogue mode: the setter's own filter designs, the chain restarts (qrl_demod_set_filter_width)"""
        _check(self.lib.qrl_demod_set_filter_width(self.h, int(width)), "qrl_demod_set_filter_width")

    def set_gain(self, value):
        """gr_demod_base::set_gain: the SSB receivers' IF gain (qrl_demod_set_gain)"""
        _check(self.lib.qrl_demod_set_gain(self.h, C.c_float(value)), "qrl_demod_set_gain")

    def stream_wait(self, hip_stream):
        """the given HIP stream (int handle) waits, on the device, for everything this handle has queued so far (qrl_demod_stream_wait)"""
        _check(self.lib.qrl_demod_stream_wait(self.h, C.c_void_p(hip_stream)), "qrl_demod_stream_wait")

    @property
    def stream(self):
        """the handle's main HIP stream (int): the one the front end of a call is launched on (qrl_demod_stream)"""
        return int(self.lib.qrl_demod_stream(self.h) or 0)

    @property
    def internal_streams(self):
        """the HIP streams (ints, without duplicates / NULLs) a call's stages are launched on: profiling aid (qrl_demod_internal_streams)"""
        arr = (C.c_void_p * 3)()
        _check(self.lib.qrl_demod_internal_streams(self.h, arr), "qrl_demod_internal_streams")
        return list(dict.fromkeys(int(x) for x in arr if x))

    def profile(self, enable=True):
        _check(self.lib.qrl_demod_profile(self.h, int(enable)), "qrl_demod_profile")

    def profile_read(self):
        ms, n, name = C.c_double(), C.c_uint64(), C.c_char_p()
        _check(self.lib.qrl_demod_profile_read(self.h, C.byref(ms), C.byref(n), C.byref(name)), "qrl_demod_profile_read")
        return ms.value, n.value, name.value.decode()

    def reset(self):
        _check(self.lib.qrl_demod_reset(self.h), "qrl_demod_reset")

    def enable_dmo_sink(self, cap_frames=64):
        """QRL_MODEM_DMR: gr_dmr_dmo_sink on the device; after every process() self.dmo_frames uint8 [batch, cap, 40] /
        self.dmo_counts int32 [batch] hold the DMR bursts cut from the call's samples (qrl_demod_set_dmo_output)"""
        t = self.torch
        dev = self.bits_a.device
        self.dmo_frames = t.zeros((self.batch, cap_frames, 40), dtype=t.uint8, device=dev)
        self.dmo_counts = t.zeros((self.batch,), dtype=t.int32, device=dev)
        _check(self.lib.qrl_demod_set_dmo_output(self.h, self.dmo_frames.data_ptr(), cap_frames, self.dmo_counts.data_ptr()),
               "qrl_demod_set_dmo_output")

    def dmo_records(self):
        """host copy of the last call's records: per stream a list of (type, fn, colour code, 33 bytes)"""
        f, c = self.dmo_frames.cpu().numpy(), self.dmo_counts.cpu().numpy()
        return [[(int(f[b, i, 0]), int(f[b, i, 1]), int(f[b, i, 2]), f[b, i, 4:37].tobytes()) for i in range(min(int(c[b]), f.shape[1]))] for b in range(self.batch)]

    def set_option(self, option, value):
        _check(self.lib.qrl_demod_set_option(self.h, int(option), int(value)), "qrl_demod_set_option")

    def set_carrier_offset(self, hz):
        _check(self.lib.qrl_demod_set_carrier_offset(self.h, float(hz)), "qrl_demod_set_carrier_offset")

    def close(self):
        if self.h:
            self.lib.qrl_demod_destroy(self.h)
            self.h = C.c_void_p()


class Channelizer:
    """Multi-carrier MMDVM receiver: mirrors make_gr_demod_mmdvm_multi2 (reference src/gr/gr_demod_mmdvm_multi2.cpp).
    process(iq) takes complex64 cuda [batch, n] (n multiple of num_channels) and returns (int16 cuda
    [batch, channel_count, cap], counts int32 [batch, channel_count])."""

    def __init__(self, ctx, num_channels, batch, max_chunk, channel_first=0, channel_count=0, stream=None, form=0,
                 channel_separation=0, decimation=0, filter_width=0, _handle=None):
        import torch
        self.torch = torch
        self.ctx, self.lib = ctx, ctx.lib
        self.owns = _handle is None        # _handle: a qrl_chan owned by someone else (qradiolink_amd.sharding.Cluster's handles)
        if _handle is None:
            cfg = _ChanConfig()
            cfg.num_channels, cfg.channel_first, cfg.channel_count = num_channels, channel_first, channel_count
            cfg.batch, cfg.max_chunk, cfg.hip_stream = batch, max_chunk, stream
            cfg.form, cfg.channel_separation, cfg.decimation, cfg.filter_width = form, channel_separation, decimation, filter_width
            self.h = C.c_void_p()
            _check(self.lib.qrl_chan_create(ctx.h, C.byref(cfg), C.byref(self.h)), "qrl_chan_create")
        else:
            self.h = C.c_void_p(_handle)
        self.batch = batch
        self.cc = channel_count if channel_count > 0 else num_channels
        self.cap = self.lib.qrl_chan_out_cap(self.h, max_chunk)
        dev = "cuda:%d" % ctx.device
        self.out = torch.zeros((batch, self.cc, self.cap), dtype=torch.int16, device=dev)
        self.counts = torch.zeros((batch, self.cc), dtype=torch.int32, device=dev)
        # rssi_tag_block outputs: one dB value per 300 samples of each 24 ksps channel
        self.rssi_cap = self.cap // 300 + 2
        self.rssi = torch.zeros((batch, self.cc, self.rssi_cap), dtype=torch.float32, device=dev)
        self.rssi_counts = torch.zeros((batch, self.cc), dtype=torch.int32, device=dev)
        _check(self.lib.qrl_chan_set_rssi_output(self.h, self.rssi.data_ptr(), self.rssi_cap, self.rssi_counts.data_ptr()),
               "qrl_chan_set_rssi_output")

    def enable_4fsk(self):
        """4FSK symbol tail (gr_demod_dmr chain) behind every channel: self.dibits uint8 [batch, cc, cap], self.fsk_counts [batch, cc, 4]"""
        t = self.torch
        dev = self.out.device
        self.fsk_cap = 2 * (self.cap // 4 + 16)
        self.dibits = t.zeros((self.batch, self.cc, self.fsk_cap), dtype=t.uint8, device=dev)
        self.fsk_const = t.zeros((self.batch, self.cc, self.fsk_cap // 2), dtype=t.complex64, device=dev)
        self.fsk_counts = t.zeros((self.batch, self.cc, 4), dtype=t.int32, device=dev)
        _check(self.lib.qrl_chan_set_4fsk_output(self.h, self.dibits.data_ptr(), self.fsk_cap, self.fsk_const.data_ptr(), self.fsk_cap // 2,
                                                 self.fsk_counts.data_ptr()), "qrl_chan_set_4fsk_output")

    def calibrate_rssi(self, level):
        _check(self.lib.qrl_chan_calibrate_rssi(self.h, float(level)), "qrl_chan_calibrate_rssi")

    def reset(self):
        """back to the state of a new handle (history, rings, symbol-sync loops): qrl_chan_reset"""
        _check(self.lib.qrl_chan_reset(self.h), "qrl_chan_reset")

    def process_async(self, iq):
        assert iq.is_cuda and iq.dtype == self.torch.complex64 and iq.dim() == 2 and iq.shape[0] == self.batch and iq.stride(1) == 1
        self.torch.cuda.current_stream().synchronize()
        _check(self.lib.qrl_chan_process(self.h, iq.data_ptr(), iq.stride(0), iq.shape[1], self.out.data_ptr(), self.cap,
                                         self.counts.data_ptr()), "qrl_chan_process")

    def sync(self):
        _check(self.lib.qrl_chan_sync(self.h), "qrl_chan_sync")

    def channelize_async(self, iq, chan_out, groups):
        """PFB only (qrl_chan_channelize): chan_out complex64 cuda [groups, batch, channel_count // groups, pitch].
        Unlike process_async this does NOT wait for torch's stream on the host: order the handle's stream behind whatever produced
        iq / last read chan_out with wait_for(stream), and the consumer behind this call with stream_wait(stream)."""
        assert iq.is_cuda and iq.dtype == self.torch.complex64 and iq.dim() == 2 and iq.shape[0] == self.batch and iq.stride(1) == 1
        assert chan_out.is_cuda and chan_out.dtype == self.torch.complex64 and chan_out.is_contiguous()
        assert tuple(chan_out.shape[:3]) == (groups, self.batch, self.cc // groups)
        _check(self.lib.qrl_chan_channelize(self.h, iq.data_ptr(), iq.stride(0), iq.shape[1], chan_out.data_ptr(), chan_out.shape[3], groups),
               "qrl_chan_channelize")

    def process_channels_async(self, chan_in, n1):
        """form 3 handle: the per-channel chain on chan_in complex64 cuda [batch, pitch] (n1 valid items per row).
        No host synchronisation with torch's stream (see channelize_async): call wait_for(stream) first when chan_in was produced there
        (a collective, a copy)."""
        assert chan_in.is_cuda and chan_in.dtype == self.torch.complex64 and chan_in.dim() == 2 and chan_in.shape[0] == self.batch and chan_in.stride(1) == 1
        _check(self.lib.qrl_chan_process_channels(self.h, chan_in.data_ptr(), chan_in.stride(0), n1, self.out.data_ptr(), self.cap,
                                                  self.counts.data_ptr()), "qrl_chan_process_channels")

    def wait_for(self, hip_stream):
        """this handle's stream waits (on the device) for what the given HIP stream (int handle) has queued so far"""
        _check(self.lib.qrl_chan_wait_for(self.h, C.c_void_p(hip_stream)), "qrl_chan_wait_for")

    def set_option(self, option, value):
        _check(self.lib.qrl_chan_set_option(self.h, int(option), int(value)), "qrl_chan_set_option")

    def profile(self, enable=True):
        _check(self.lib.qrl_chan_profile(self.h, int(enable)), "qrl_chan_profile")

    def profile_read(self):
        ms, n, name = C.c_double(), C.c_uint64(), C.c_char_p()
        _check(self.lib.qrl_chan_profile_read(self.h, C.byref(ms), C.byref(n), C.byref(name)), "qrl_chan_profile_read")
        return ms.value, n.value, name.value.decode()

    def profile_read_kernels(self):
        """[(kernel, total ms, launches)] of the channelizer, the fused per-channel kernel and the symbol synchroniser (qrl_chan_profile_read_kernels)"""
        ms, n = (C.c_double * 3)(), (C.c_uint64 * 3)()
        _check(self.lib.qrl_chan_profile_read_kernels(self.h, ms, n), "qrl_chan_profile_read_kernels")
        return [(k, ms[i], n[i]) for i, k in enumerate(("channelizer", "k_chan_tail", "k_symsync_ff"))]

    def stream_wait(self, hip_stream):
        """the given HIP stream (int handle, e.g. torch.cuda.current_stream().cuda_stream) waits on the device for this handle's work so far"""
        _check(self.lib.qrl_chan_stream_wait(self.h, C.c_void_p(hip_stream)), "qrl_chan_stream_wait")

    @property
    def internal_streams(self):
        """the HIP streams (ints) a call's stages are launched on: profiling aid (qrl_chan_internal_streams)"""
        arr = (C.c_void_p * 3)()
        _check(self.lib.qrl_chan_internal_streams(self.h, arr), "qrl_chan_internal_streams")
        return list(dict.fromkeys(int(x) for x in arr if x))

    def process(self, iq):
        self.process_async(iq)
        self.sync()
        return self.out, self.counts

    def close(self):
        if self.h:
            if self.owns:
                self.lib.qrl_chan_destroy(self.h)
            self.h = C.c_void_p()


class Deframer:
    """gr_deframer_bb on the device (reference src/gr/gr_deframer_bb.cpp): process(bits, counts) takes the uint8 cuda tensor
    [batch, cap] of one demodulator port plus its per-stream valid counts (int32 cuda [batch] view or None) and returns
    (uint8 cuda [batch, out_cap], int32 cuda [batch]) = what the block pushes into its mailbox."""

    def __init__(self, ctx, deframer_type, batch, stream=None):
        import torch
        self.torch = torch
        self.ctx, self.lib, self.batch = ctx, ctx.lib, batch
        self.h = C.c_void_p()
        _check(self.lib.qrl_deframer_create(ctx.h, deframer_type, batch, stream, C.byref(self.h)), "qrl_deframer_create")
        self.out = None
        self.out_counts = torch.zeros((batch,), dtype=torch.int32, device="cuda:%d" % ctx.device)

    def process(self, bits, counts=None, count_stride=1, n=None):
        t = self.torch
        assert bits.is_cuda and bits.dtype == t.uint8 and bits.dim() == 2 and bits.shape[0] == self.batch and bits.stride(1) == 1
        n = bits.shape[1] if n is None else n
        cap = 2 * n + 24
        if self.out is None or self.out.shape[1] < cap:
            self.out = t.zeros((self.batch, cap), dtype=t.uint8, device=bits.device)
        t.cuda.current_stream().synchronize()   # the handle has its own HIP stream: the producer of `bits` must be done
        _check(self.lib.qrl_deframer_process(self.h, bits.data_ptr(), bits.stride(0), n, counts.data_ptr() if counts is not None else None,
                                             count_stride, self.out.data_ptr(), self.out.shape[1], self.out_counts.data_ptr()),
               "qrl_deframer_process")
        _check(self.lib.qrl_deframer_sync(self.h), "qrl_deframer_sync")
        return self.out, self.out_counts

    def reset(self):
        _check(self.lib.qrl_deframer_reset(self.h), "qrl_deframer_reset")

    def close(self):
        if self.h:
            self.lib.qrl_deframer_destroy(self.h)
            self.h = C.c_void_p()


class Rssi:
    """rssi_block on the device (reference src/gr/rssi_block.cpp:25-50): process(filtered, counts) takes the complex64 cuda tensor
    [batch, cap] of demodulator port 0 plus the per-stream item counts (int32 cuda view, stride in elements, or None) and returns
    (float32 cuda [batch, cap] dB values, float32 cuda [batch] latest value = probe_signal_f::level())."""

    def __init__(self, ctx, batch, level=0.0, stream=None):
        import torch
        self.torch = torch
        self.ctx, self.lib, self.batch = ctx, ctx.lib, batch
        self.h = C.c_void_p()
        _check(self.lib.qrl_rssi_create(ctx.h, batch, level, stream, C.byref(self.h)), "qrl_rssi_create")
        self.last = torch.zeros((batch,), dtype=torch.float32, device="cuda:%d" % ctx.device)
        self.out_counts = torch.zeros((batch,), dtype=torch.int32, device="cuda:%d" % ctx.device)
        self.out = None

    def set_level(self, level):
        _check(self.lib.qrl_rssi_set_level(self.h, level), "qrl_rssi_set_level")

    def process(self, filtered, counts=None, count_stride=1, n=None):
        t = self.torch
        assert filtered.is_cuda and filtered.dtype == t.complex64 and filtered.dim() == 2 and filtered.shape[0] == self.batch and filtered.stride(1) == 1
        n = filtered.shape[1] if n is None else n
        if self.out is None or self.out.shape[1] < n:
            self.out = t.zeros((self.batch, max(n, 1)), dtype=t.float32, device=filtered.device)
        t.cuda.current_stream().synchronize()
        _check(self.lib.qrl_rssi_process(self.h, filtered.data_ptr(), filtered.stride(0), n, counts.data_ptr() if counts is not None else None,
                                         count_stride, self.out.data_ptr(), self.out.shape[1], self.last.data_ptr(), self.out_counts.data_ptr()),
               "qrl_rssi_process")
        _check(self.lib.qrl_rssi_sync(self.h), "qrl_rssi_sync")
        return self.out, self.last

    def reset(self):
        _check(self.lib.qrl_rssi_reset(self.h), "qrl_rssi_reset")

    def close(self):
        if self.h:
            self.lib.qrl_rssi_destroy(self.h)
            self.h = C.c_void_p()


class Fft:
    """rx_fft_c on the device (reference src/gr/rx_fft.cpp:44-213): work(iq) feeds complex64 cuda [batch, n]; get_fft_data()
    returns float32 cuda [batch, fftsize] (dB, negative frequencies first) or None while no spectrum is ready."""

    def __init__(self, ctx, batch, fftsize=32768, wintype=5, stream=None):
        import torch
        self.torch = torch
        self.ctx, self.lib, self.batch = ctx, ctx.lib, batch
        self.h = C.c_void_p()
        _check(self.lib.qrl_fft_create(ctx.h, batch, fftsize, wintype, stream, C.byref(self.h)), "qrl_fft_create")

    def set_enabled(self, enabled):
        _check(self.lib.qrl_fft_set_enabled(self.h, int(bool(enabled))), "qrl_fft_set_enabled")

    def set_fft_size(self, n):
        _check(self.lib.qrl_fft_set_fft_size(self.h, n), "qrl_fft_set_fft_size")

    def get_fft_size(self):
        return int(self.lib.qrl_fft_get_fft_size(self.h))

    def set_window_type(self, w):
        _check(self.lib.qrl_fft_set_window_type(self.h, w), "qrl_fft_set_window_type")

    def get_window_type(self):
        return int(self.lib.qrl_fft_get_window_type(self.h))

    def work(self, iq):
        t = self.torch
        assert iq.is_cuda and iq.dtype == t.complex64 and iq.dim() == 2 and iq.shape[0] == self.batch and iq.stride(1) == 1
        t.cuda.current_stream().synchronize()
        _check(self.lib.qrl_fft_process(self.h, iq.data_ptr(), iq.stride(0), iq.shape[1]), "qrl_fft_process")
        _check(self.lib.qrl_fft_sync(self.h), "qrl_fft_sync")   # the handle's stream has read `iq`: the caller may free / reuse it

    def get_fft_data(self):
        t = self.torch
        n = self.get_fft_size()
        out = t.empty((self.batch, n), dtype=t.float32, device="cuda:%d" % self.ctx.device)
        got = C.c_uint(0)
        _check(self.lib.qrl_fft_get_fft_data(self.h, out.data_ptr(), out.stride(0), C.byref(got)), "qrl_fft_get_fft_data")
        _check(self.lib.qrl_fft_sync(self.h), "qrl_fft_sync")
        return out if got.value else None

    def close(self):
        if self.h:
            self.lib.qrl_fft_destroy(self.h)
            self.h = C.c_void_p()


class Synth:
    """Multi-carrier MMDVM transmitter: mirrors make_gr_mod_mmdvm_multi2 (reference src/gr/gr_mod_mmdvm_multi2.cpp).
    process(x) takes int16 cuda [batch, num_channels, n] (24 ksps FM baseband per channel) and returns complex64 cuda
    [batch, produced] at 250 ksps."""

    def __init__(self, ctx, num_channels, batch, max_samples, filter_width=0, stream=None, bb_gain=1.0, single_carrier=False):
        import torch
        self.torch = torch
        self.ctx, self.lib, self.batch, self.nch = ctx, ctx.lib, batch, num_channels
        cfg = _SynthConfig()
        cfg.single_carrier = 1 if single_carrier else 0
        cfg.num_channels, cfg.filter_width, cfg.batch, cfg.max_samples = num_channels, filter_width, batch, max_samples
        cfg.hip_stream, cfg.bb_gain = stream, bb_gain
        self.h = C.c_void_p()
        _check(self.lib.qrl_synth_create(ctx.h, C.byref(cfg), C.byref(self.h)), "qrl_synth_create")

    def process(self, x):
        t = self.torch
        assert x.is_cuda and x.dtype == t.int16 and x.dim() == 3 and x.shape[0] == self.batch and x.shape[1] == self.nch
        x = x.contiguous()
        n = x.shape[2]
        cap = self.lib.qrl_synth_out_cap(self.h, n)
        out = t.zeros((self.batch, cap), dtype=t.complex64, device=x.device)
        produced = C.c_size_t(0)
        t.cuda.current_stream().synchronize()
        _check(self.lib.qrl_synth_process(self.h, x.data_ptr(), n, n, out.data_ptr(), cap, C.byref(produced)), "qrl_synth_process")
        _check(self.lib.qrl_synth_sync(self.h), "qrl_synth_sync")
        return out[:, :produced.value]

    def add_zero_runs(self, runs):
        """gr_zero_idle_bursts: runs = [(stream, channel, start, count), ...] at the rate of that block's input (qrl_synth_add_zero_runs)"""
        arr = (_ZeroRun * len(runs))(*[_ZeroRun(*r) for r in runs])
        _check(self.lib.qrl_synth_add_zero_runs(self.h, arr, len(runs)), "qrl_synth_add_zero_runs")

    def close(self):
        if self.h:
            self.lib.qrl_synth_destroy(self.h)
            self.h = C.c_void_p()


class FrameSync:
    """gr_modem::synchronize/findSync/packBytes on the device (reference src/gr_modem.cpp:1119-1282): process(bits, counts) returns
    (uint8 cuda [batch, out_cap] records {u32 frame_type, u32 nbytes, payload padded to 4}, int32 cuda [batch, 2] = bytes, frames)."""

    def __init__(self, ctx, modem_type, batch, stream=None):
        import torch
        self.torch = torch
        self.ctx, self.lib, self.batch = ctx, ctx.lib, batch
        self.h = C.c_void_p()
        _check(self.lib.qrl_framesync_create(ctx.h, modem_type, batch, stream, C.byref(self.h)), "qrl_framesync_create")
        self.frame_bytes = self.lib.qrl_framesync_frame_bytes(self.h)
        self.out = None
        self.out_counts = torch.zeros((batch, 2), dtype=torch.int32, device="cuda:%d" % ctx.device)
        # bits collected into a frame while a sync was held, per call: > 0 <=> gr_modem::synchronize's data_to_process (src/gr_modem.cpp:1121-1175)
        self.activity = torch.zeros((batch,), dtype=torch.int32, device="cuda:%d" % ctx.device)
        _check(self.lib.qrl_framesync_set_activity_output(self.h, self.activity.data_ptr()), "qrl_framesync_set_activity_output")

    def process(self, bits, counts=None, count_stride=1, n=None):
        t = self.torch
        assert bits.is_cuda and bits.dtype == t.uint8 and bits.dim() == 2 and bits.shape[0] == self.batch and bits.stride(1) == 1
        n = bits.shape[1] if n is None else n
        # worst case: a frame begun in earlier calls completes now, then back-to-back frames (8-byte header + padding each)
        cap = (n // 8 + self.frame_bytes + 80 + 16 * (n // max(8 * self.frame_bytes, 8) + 2) + 3) & ~3
        if self.out is None or self.out.shape[1] < cap:
            self.out = t.zeros((self.batch, cap), dtype=t.uint8, device=bits.device)
        t.cuda.current_stream().synchronize()   # the handle has its own HIP stream: the producer of `bits` must be done
        _check(self.lib.qrl_framesync_process(self.h, bits.data_ptr(), bits.stride(0), n, counts.data_ptr() if counts is not None else None,
                                              count_stride, self.out.data_ptr(), self.out.shape[1], self.out_counts.data_ptr()),
               "qrl_framesync_process")
        _check(self.lib.qrl_framesync_sync(self.h), "qrl_framesync_sync")
        return self.out, self.out_counts

    def reset(self):
        _check(self.lib.qrl_framesync_reset(self.h), "qrl_framesync_reset")

    def close(self):
        if self.h:
            self.lib.qrl_framesync_destroy(self.h)
            self.h = C.c_void_p()


class Mod:
    """Batch TX modulator: mirrors make_gr_mod_qpsk (reference src/gr/gr_mod_qpsk.cpp:19-30).
    process(bytes) takes a torch cuda uint8 tensor [batch, nbytes] (packed bytes, as gr_byte_source hands them
    out) and returns a complex64 cuda tensor [batch, nbytes * 8 * sps]."""

    def __init__(self, ctx, modem_type, batch, max_bytes, stream=None, bb_gain=1.0, device_samp_rate=0,
                 carrier_offset_hz=0.0):
        import torch
        self.torch = torch
        self.ctx, self.lib = ctx, ctx.lib
        cfg = _ModConfig()
        cfg.modem_type = modem_type
        cfg.use_mode_defaults = 1
        cfg.batch = batch
        cfg.max_bytes = max_bytes
        cfg.hip_stream = stream
        cfg.bb_gain = bb_gain
        cfg.device_samp_rate = device_samp_rate      # gr_mod_base::set_samp_rate (back-end interpolator)
        cfg.carrier_offset_hz = carrier_offset_hz    # gr_mod_base::set_carrier_offset (rotator at 1 Msps)
        self.batch, self.max_bytes = batch, max_bytes
        self.h = C.c_void_p()
        _check(self.lib.qrl_mod_create(ctx.h, C.byref(cfg), C.byref(self.h)), "qrl_mod_create")
        self.spb = self.lib.qrl_mod_samples_per_byte(self.h)
        bpb = C.c_size_t()
        self.spblock = self.lib.qrl_mod_samples_per_block(self.h, C.byref(bpb))   # M17: 2500 samples per 3 bytes (spb = 0)
        self.bytes_per_block = bpb.value

    def process_async(self, data, out=None):
        assert data.is_cuda and data.dtype == self.torch.uint8 and data.dim() == 2 and data.shape[0] == self.batch
        assert data.stride(1) == 1
        n = data.shape[1]
        if out is None:
            out = self.torch.empty((self.batch, n // self.bytes_per_block * self.spblock), dtype=self.torch.complex64, device=data.device)
        self.torch.cuda.current_stream().synchronize()
        _check(self.lib.qrl_mod_process(self.h, data.data_ptr(), data.stride(0), n, out.data_ptr(), out.stride(0)),
               "qrl_mod_process")
        return out

    def sync(self):
        _check(self.lib.qrl_mod_sync(self.h), "qrl_mod_sync")

    def process(self, data):
        out = self.process_async(data)
        self.sync()
        return out

    def reset(self):
        _check(self.lib.qrl_mod_reset(self.h), "qrl_mod_reset")

    def set_bb_gain(self, g):
        _check(self.lib.qrl_mod_set_bb_gain(self.h, float(g)), "qrl_mod_set_bb_gain")

    def set_carrier_offset(self, hz):
        _check(self.lib.qrl_mod_set_carrier_offset(self.h, float(hz)), "qrl_mod_set_carrier_offset")

    def add_zero_runs(self, runs):
        """QRL_MODEM_DMR: the "zero_samples" tags of gr_zero_idle_bursts, runs = [(stream, T, count), ...] with T in the block's 24 ksps input
        coordinates (qrl_mod_add_zero_runs)"""
        arr = (_ZeroRun * len(runs))(*[_ZeroRun(s_, 0, t, c) for s_, t, c in runs])
        _check(self.lib.qrl_mod_add_zero_runs(self.h, arr, len(runs)), "qrl_mod_add_zero_runs")

    def close(self):
        if self.h:
            self.lib.qrl_mod_destroy(self.h)
            self.h = C.c_void_p()


class AMod:
    """Batch analogue voice modulator: mirrors make_gr_mod_nbfm (src/gr/gr_mod_nbfm.cpp:19-77) and make_gr_mod_ssb (gr_mod_ssb.cpp:19-82).
    process(audio) takes a float32 cuda tensor [batch, n] at 8 ksps (NBFM: n a multiple of 4) and returns complex64 at 1 Msps:
    [batch, 125 n] for NBFM, 125 x the audio items of the 1024-chunks the call completed for SSB."""

    def __init__(self, ctx, modem_type, batch, max_samples, bb_gain=1.0, device_samp_rate=0, carrier_offset_hz=0.0):
        import torch
        self.torch, self.ctx, self.lib = torch, ctx, ctx.lib
        cfg = _AModConfig(modem_type, batch, max_samples, None, bb_gain, device_samp_rate, carrier_offset_hz)
        self.h = C.c_void_p()
        _check(self.lib.qrl_amod_create(ctx.h, C.byref(cfg), C.byref(self.h)), "qrl_amod_create")
        self.batch, self.spa = batch, self.lib.qrl_amod_samples_per_sample(self.h)

    def process(self, audio):
        t = self.torch
        assert audio.is_cuda and audio.dtype == t.float32 and audio.dim() == 2 and audio.shape[0] == self.batch and audio.stride(1) == 1
        n = audio.shape[1]
        out = t.zeros((self.batch, max(self.lib.qrl_amod_out_cap(self.h, n), 1)), dtype=t.complex64, device=audio.device)
        t.cuda.current_stream().synchronize()
        _check(self.lib.qrl_amod_process(self.h, audio.data_ptr(), audio.stride(0), n, out.data_ptr(), out.stride(0)), "qrl_amod_process")
        _check(self.lib.qrl_amod_sync(self.h), "qrl_amod_sync")
        return out[:, :self.lib.qrl_amod_last_count(self.h)]     # SSB: whole chunks of 1024 audio items only (the cessb stretcher)

    def set_bb_gain(self, g):
        _check(self.lib.qrl_amod_set_bb_gain(self.h, C.c_float(g)), "qrl_amod_set_bb_gain")

    def set_carrier_offset(self, hz):
        _check(self.lib.qrl_amod_set_carrier_offset(self.h, C.c_double(hz)), "qrl_amod_set_carrier_offset")

    def set_cw_k(self, key_down):
        """gr_mod_base::set_cw_k: amplitude of the CW tone source, 0.98 (key down) / 0.001 (qrl_amod_set_cw_k; QRL_MODEM_CW600USB handles)"""
        _check(self.lib.qrl_amod_set_cw_k(self.h, int(bool(key_down))), "qrl_amod_set_cw_k")

    def process_cw(self, n):
        """QRL_MODEM_CW600USB: what n samples of the key's tone source give (whole chunks of 1024 like every SSB handle)"""
        t = self.torch
        out = t.zeros((self.batch, max(self.lib.qrl_amod_out_cap(self.h, n), 1)), dtype=t.complex64, device="cuda")
        t.cuda.current_stream().synchronize()
        _check(self.lib.qrl_amod_process(self.h, None, 0, n, out.data_ptr(), out.stride(0)), "qrl_amod_process")
        _check(self.lib.qrl_amod_sync(self.h), "qrl_amod_sync")
        return out[:, :self.lib.qrl_amod_last_count(self.h)]

    def set_ctcss(self, tone_hz):
        """gr_mod_nbfm::set_ctcss: tone (Hz) added to the audio, band-pass audio filter; 0 switches it off again (qrl_amod_set_ctcss)"""
        _check(self.lib.qrl_amod_set_ctcss(self.h, C.c_float(tone_hz)), "qrl_amod_set_ctcss")

    def set_filter_width(self, width):
        """gr_mod_base::set_filter_width for this handle's mode: the setter's own filter designs, the chain restarts (qrl_amod_set_filter_width)"""
        _check(self.lib.qrl_amod_set_filter_width(self.h, int(width)), "qrl_amod_set_filter_width")

    def reset(self):
        _check(self.lib.qrl_amod_reset(self.h), "qrl_amod_reset")

    def close(self):
        if self.h:
            self.lib.qrl_amod_destroy(self.h)
            self.h = None


def bptc19696_decode(ctx, bursts):
    """[n, 33] uint8 cuda tensor of DMR bursts -> [n, 12] payloads (CBPTC19696::decode)"""
    import torch
    assert bursts.is_cuda and bursts.dtype == torch.uint8 and bursts.dim() == 2 and bursts.shape[1] == 33 and bursts.is_contiguous()
    out = torch.zeros((bursts.shape[0], 12), dtype=torch.uint8, device=bursts.device)
    torch.cuda.current_stream().synchronize()
    _check(ctx.lib.qrl_bptc19696_decode(ctx.h, None, bursts.data_ptr(), bursts.shape[0], out.data_ptr()), "qrl_bptc19696_decode")
    torch.cuda.synchronize()
    return out


def bptc19696_encode(ctx, payloads, bursts):
    """[n, 12] payloads written into the code bits of [n, 33] bursts in place (CBPTC19696::encode); returns bursts"""
    import torch
    assert payloads.is_cuda and bursts.is_cuda and payloads.shape[1] == 12 and bursts.shape[1] == 33 and payloads.shape[0] == bursts.shape[0]
    assert payloads.is_contiguous() and bursts.is_contiguous() and payloads.dtype == torch.uint8 and bursts.dtype == torch.uint8
    torch.cuda.current_stream().synchronize()
    _check(ctx.lib.qrl_bptc19696_encode(ctx.h, None, payloads.data_ptr(), payloads.shape[0], bursts.data_ptr()), "qrl_bptc19696_encode")
    torch.cuda.synchronize()
    return bursts


def m17_decode_frames(ctx, frames):
    """[n, 48] uint8 cuda tensor of M17 frames -> [n, 40] records (qrl_m17_decode_frames)"""
    import torch
    assert frames.is_cuda and frames.dtype == torch.uint8 and frames.dim() == 2 and frames.shape[1] == 48 and frames.is_contiguous()
    out = torch.zeros((frames.shape[0], 40), dtype=torch.uint8, device=frames.device)
    torch.cuda.current_stream().synchronize()
    _check(ctx.lib.qrl_m17_decode_frames(ctx.h, None, frames.data_ptr(), frames.shape[0], out.data_ptr()), "qrl_m17_decode_frames")
    torch.cuda.synchronize()
    return out


def m17_encode_frames(ctx, records):
    """[n, 40] uint8 cuda tensor of records (layout of m17_decode_frames) -> [n, 48] frames (qrl_m17_encode_frames)"""
    import torch
    assert records.is_cuda and records.dtype == torch.uint8 and records.dim() == 2 and records.shape[1] == 40 and records.is_contiguous()
    out = torch.zeros((records.shape[0], 48), dtype=torch.uint8, device=records.device)
    torch.cuda.current_stream().synchronize()
    _check(ctx.lib.qrl_m17_encode_frames(ctx.h, None, records.data_ptr(), records.shape[0], out.data_ptr()), "qrl_m17_encode_frames")
    torch.cuda.synchronize()
    return out


def collect(dem, iq, chunk):
    """Run a whole [B, N] device tensor through dem in calls of `chunk` samples and concatenate the
    per-port outputs on the host: returns dict of lists (one numpy array per stream)."""
    import torch
    B, N = iq.shape
    ports = {k: [[] for _ in range(B)] for k in ("filtered", "constellation", "bits_a", "bits_b")}
    idx = {"filtered": 0, "constellation": 1, "bits_a": 2, "bits_b": 3}
    if getattr(dem, "audio_cap", 0):   # analogue voice receivers: port 1 carries audio
        ports = {k: [[] for _ in range(B)] for k in ("filtered", "audio")}
        idx = {"filtered": 0, "audio": 1}
    for s in range(0, N, chunk):
        part = iq[:, s:s + chunk]
        if part.stride(0) % 2 or (part.data_ptr() % 16):
            part = part.contiguous()
            if part.stride(0) % 2:
                pad = torch.zeros((B, part.shape[1] + 1), dtype=part.dtype, device=part.device)
                pad[:, :part.shape[1]] = part
                part = pad[:, :part.shape[1]]
        out = dem.process(part)
        cnt = out["counts"].cpu().numpy()
        for k, j in idx.items():
            if out[k] is None:
                continue
            host = out[k].cpu().numpy()
            for b in range(B):
                ports[k][b].append(host[b, :cnt[b, j]].copy())
    return {k: [np.concatenate(v) if v else np.zeros(0) for v in ports[k]] for k in ports}
