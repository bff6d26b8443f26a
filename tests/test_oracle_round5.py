"""Round 5 oracle additions on CPU: analog::sig_source_f as restated for the TX-side CTCSS of gr_mod_nbfm ([GR-MEM]: fixed-point NCO + 1024-row
sine table), the NBFM modulator with set_ctcss, gr_mod_dmr with its zero-idle block."""
import ctypes as C

import numpy as np

import orc


def test_sig_source_cos_is_a_cosine_to_the_table_accuracy():
    for fs, f, amp in ((8000.0, 88.5, 0.15), (8000.0, 81.5, 0.15), (8000.0, 250.3, 1.0)):
        y = orc.sig_source_cos(fs, f, amp, 16000)
        inc = orc.lib.orc_fxpt_phase_inc(C.c_double(fs), C.c_double(f))
        # the NCO's own frequency: inc / 2^32 cycles per sample (the float conversion of 2 pi f / fs is truncated to the fixed-point grid)
        k = np.arange(y.size, dtype=np.float64)
        ref = amp * np.cos(2 * np.pi * ((k * inc) % 2 ** 32) / 2 ** 32)
        assert np.max(np.abs(y - ref)) < 3e-6 * max(amp, 1.0) + 2e-7          # piecewise-linear table, 1024 rows: ~ (pi / 1024)^2 / 8 / 2 of the amplitude
        assert abs(inc / 2 ** 32 * fs - f) < 1e-4                             # within the float grid of the angle rate
        assert y[0] == np.float32(np.float64(np.float32(y[0] / amp)) * amp) or abs(y[0] - amp) < 1e-6


def test_sig_source_cos_is_index_addressed():
    a = orc.sig_source_cos(8000, 88.5, 0.15, 5000)
    b = orc.sig_source_cos(8000, 88.5, 0.15, 3000, k0=2000)
    assert np.array_equal(a[2000:], b)


def test_nbfm_modulator_ctcss_variants():
    n = 8000
    audio = (0.5 * np.sin(2 * np.pi * 700 * np.arange(n) / 8000.0)).astype(np.float32)
    plain = orc.mod_nbfm(audio)
    tone = orc.mod_nbfm(audio, ctcss=88.5)
    off = orc.mod_nbfm(audio, ctcss=-1.0)
    assert plain.size == tone.size == off.size == 125 * n
    assert not np.array_equal(plain, tone) and not np.array_equal(plain, off)
    # the sub-tone is in the instantaneous frequency of the tone variant only: an 88.5 Hz line
    def line(x, f):
        ph = np.unwrap(np.angle(x[100000:900000:25].astype(np.complex128)))       # 40 ksps
        d = np.diff(ph)
        s = np.abs(np.fft.rfft((d - d.mean()) * np.hanning(d.size)))
        k = int(round(f * d.size / 40000.0))
        return s[k - 2:k + 3].max() / np.median(s[5:400])
    assert line(tone, 88.5) > 50 and line(plain, 88.5) < 10


def test_mod_dmr_shape_and_zero_runs():
    data = (np.arange(198) * 37 % 251).astype(np.uint8)
    y = orc.mod_dmr(data)
    assert y.size == 198 // 3 * 2500
    assert not y[:1439 * 125 // 3 - 2000].any()                                # the zero-idle block's history: 1439 items at 24 ksps of silence first
    z = orc.mod_dmr(data, zero_runs=[(1439 + 62 + 100, 300)])
    lo, hi = (1439 + 100 + 60) * 125 // 3, (1439 + 400 - 60) * 125 // 3
    assert np.abs(z[lo:hi]).max() < 1e-3 and np.abs(y[lo:hi]).min() > 0.3
    assert np.array_equal(z[:lo - 6000], y[:lo - 6000]) and np.array_equal(z[hi + 6000:], y[hi + 6000:])


def test_ctcss_guard_tones_known_answers():
    """ctcss_squelch_ff's neighbours: the table entries either side of a standard tone, freq * 0.98 / freq * 1.02 (double literals, narrowed to float)
    for an off-table tone -- the reference's tone list has two such entries, 81.5 and 87.4 Hz (src/ext/utils.h:17)"""
    def guards(f):
        fl, fr = C.c_float(), C.c_float()
        orc.lib.orc_ctcss_freqs(C.c_float(f), C.byref(fl), C.byref(fr))
        return fl.value, fr.value
    assert guards(88.5) == (np.float32(85.4), np.float32(91.5))
    assert guards(67.0) == (np.float32(np.float64(np.float32(67.0)) * 0.98), np.float32(71.9))
    for f in (81.5, 87.4):
        assert guards(f) == (np.float32(np.float64(np.float32(f)) * 0.98), np.float32(np.float64(np.float32(f)) * 1.02))


def test_sig_source_sin_with_offset_and_the_cw_branch_of_the_reference():
    """the CW key's tone source: sig_source_f(8000, GR_SIN_WAVE, 600, 0.001, 1) into gr_mod_ssb(125, 1000000, 1700, 1000, 0); set_cw_k switches the amplitude
    between 0.98 and 0.001 (src/gr/gr_mod_base.cpp:144,180,679-683,948-956 -- checked against the source text where the reference is present)"""
    import os
    for amp in (0.98, 0.001):
        y = orc.sig_source_sin(8000, 600, amp, 16000, offset=1.0)
        inc = orc.lib.orc_fxpt_phase_inc(C.c_double(8000.0), C.c_double(600.0))
        k = np.arange(y.size, dtype=np.float64)
        ref = amp * np.sin(2 * np.pi * ((k * inc) % 2 ** 32) / 2 ** 32) + 1.0
        assert np.max(np.abs(y - ref)) < 3e-6 + 2e-7
    a = orc.sig_source_sin(8000, 600, 0.98, 5000, offset=1.0)
    b = orc.sig_source_sin(8000, 600, 0.98, 3000, k0=2000, offset=1.0)
    assert np.array_equal(a[2000:], b) and a[0] == 1.0
    # the quarter turn between the two waveforms of the source
    c = orc.sig_source_cos(8000, 1000, 1.0, 64)
    s = orc.sig_source_sin(8000, 1000, 1.0, 64, k0=2)                  # 1000 Hz at 8 ksps: two samples = a quarter turn
    assert np.max(np.abs(c[:62] - s[:62])) < 1e-5
    src = "/root/reference/src/gr/gr_mod_base.cpp"
    if os.path.exists(src):
        text = open(src).read().replace(" ", "").replace("\n", "")
        assert "_signal_source=gr::analog::sig_source_f::make(8000,gr::analog::GR_SIN_WAVE,600,0.001,1);" in text
        assert "_usb_cw=make_gr_mod_ssb(125,1000000,1700,1000,0);" in text
        assert "_top_block->connect(_signal_source,0,_usb_cw,0);_top_block->connect(_usb_cw,0,_rotator,0);" in text
        assert "if(value)a=0.98;elsea=0.001;_signal_source->set_amplitude(a);" in text


def test_scope_tap_setters_of_the_reference_source_text():
    """what qrl_demod_config.time_domain_samp_rate / time_domain_filter_width restate (src/gr/gr_demod_base.cpp:1249-1301), checked against the source text where
    the reference is present: integer decimation 1e6 / samp_rate, low_pass(1, 1e6, samp_rate/2 - samp_rate/8, samp_rate/4, HAMMING), and the width setter's
    low_pass(1, 1e6, width, width, HAMMING); rates above 1 Msps are ignored"""
    import os
    src = "/root/reference/src/gr/gr_demod_base.cpp"
    if not os.path.exists(src):
        return
    text = open(src).read().replace(" ", "").replace("\n", "")
    assert "if((uint)samp_rate>INTERNAL_DEFAULT_SAMPLE_RATE)return;" in text
    assert "intdecimation=INTERNAL_DEFAULT_SAMPLE_RATE/samp_rate;" in text
    assert "firdes::low_pass(1,INTERNAL_DEFAULT_SAMPLE_RATE,samp_rate/2-samp_rate/8,samp_rate/4,gr::fft::window::WIN_HAMMING);" in text
    assert "_resampler_time_domain=gr::filter::rational_resampler_ccf::make(1,decimation,taps);" in text
    assert "firdes::low_pass(1,INTERNAL_DEFAULT_SAMPLE_RATE,filter_width,filter_width,gr::fft::window::WIN_HAMMING);_resampler_time_domain->set_taps(taps);" in text
    # the integer divisions matter: 50 ksps -> cutoff 18750, transition 12500
    t = orc.low_pass(1, 1000000, 50000 // 2 - 50000 // 8, 50000 // 4)
    assert t.size % 2 == 1 and abs(float(t.sum()) - 1.0) < 1e-5
