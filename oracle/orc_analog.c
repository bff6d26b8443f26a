/* orc_analog.c -- CPU ORACLE (TEST INFRASTRUCTURE, never shipped or linked by the product): the analogue voice receivers of
 * QRadioLink's gr_demod_base, restated block by block.
 *   gr_demod_nbfm   reference src/gr/gr_demod_nbfm.cpp:31-88   (instances gr_demod_base.cpp:219-220: filter width 2500 / 5000)
 *   gr_demod_am     reference src/gr/gr_demod_am.cpp:28-79     (instance  gr_demod_base.cpp:215: filter width 5000)
 *   gr_demod_wbfm   reference src/gr/gr_demod_wbfm.cpp:28-72   (instance  gr_demod_base.cpp:228: filter width 75000)
 *   gr_demod_ssb    reference src/gr/gr_demod_ssb.cpp:28-81    (instances gr_demod_base.cpp:226-227: filter width 2700, USB / LSB)
 *   cessb clipper / stretcher   src/gr/cessb/clipper_cc_impl.cc:66-93, stretcher_cc_impl.cc:68-106 (in the reference tree)
 *   de-emphasis taps            src/gr/emphasis.cpp:16-43
 * GNU Radio 3.10 block semantics restated from memory [GR-MEM] (parity unpinned):
 *   pwr_squelch_cc / squelch_base_cc  gr-analog/lib/squelch_base_cc_impl.cc, pwr_squelch_cc_impl.cc
 *   agc2_ff                           gr-analog/include/gnuradio/analog/agc2.h
 *   iir_filter_ffd                    gr-filter/include/gnuradio/filter/iir_filter.h (double taps, double accumulator, float in/out)
 *   rational_resampler_fff, fft_filter_{ccf,ccc,fff}, quadrature_demod_cf, complex_to_mag: as in orc_blocks.c */
#include "orc.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NEW(T, n) ((T*)calloc((size_t)(n) + 16, sizeof(T)))

/* firdes::complex_band_pass_2 = low_pass_2 prototype rotated to the band centre, like complex_band_pass */
int orc_complex_band_pass_2(double gain, double fs, double lo, double hi, double tw, double atten_db, int win, cf32* taps)
{
    int ntaps = orc_low_pass_2(gain, fs, (hi - lo) / 2, tw, atten_db, win, NULL);
    if (!taps) return ntaps;
    float* lp = NEW(float, ntaps);
    orc_low_pass_2(gain, fs, (hi - lo) / 2, tw, atten_db, win, lp);
    float freq = (float)(M_PI * (hi + lo) / fs);
    float phase;
    if (ntaps & 1) phase = -freq * (float)(ntaps >> 1);
    else           phase = (float)(-freq / 2.0 * ((1 + 2 * ntaps) >> 1));
    for (int i = 0; i < ntaps; i++) {
        taps[i].re = (float)(lp[i] * cos((double)phase));
        taps[i].im = (float)(lp[i] * sin((double)phase));
        phase += freq;
    }
    free(lp);
    orc_trace_taps(taps, sizeof(cf32) * (size_t)ntaps, "complex_band_pass_2(%.17g,%.17g,%.17g,%.17g,%.17g,%.17g,%d)", gain, fs, lo, hi, tw, atten_db, win);
    return ntaps;
}

/* emphasis.cpp:16-43 (tanf on a float argument, the rest in double): b = {b0, b0}, a = {1, -p1} */
void orc_deemph_taps(int sample_rate, double tau, double a[2], double b[2])
{
    const double fs = (double)sample_rate;
    const double w_c = 1.0 / tau;
    const double w_ca = 2.0 * fs * (double)tanf((float)(w_c / (2.0 * fs)));
    const double k = -w_ca / (2.0 * fs);
    const double z1 = -1.0;
    const double p1 = (1.0 + k) / (1.0 - k);
    const double b0 = -k / (1.0 - k);
    b[0] = b0 * 1.0; b[1] = b0 * -z1;
    a[0] = 1.0;      a[1] = -p1;
}

/* squelch envelope while ramping: 0.5 - cos(pi k / ramp) / 2 in double, k = 0 .. ramp (used as a float factor) */
void orc_squelch_envelope(int ramp, float* env)
{
    for (int k = 0; k <= ramp; k++) env[k] = (float)(0.5 - cos(M_PI * (double)k / (double)ramp) / 2.0);
}

/* pwr_squelch_cc(db, alpha, ramp, gate): states MUTED 0 / ATTACK 1 / UNMUTED 2 / DECAY 3.  Per input item: the single-pole power
 * estimate (double) is updated, then the state machine steps, then -- unless MUTED -- the item times the envelope is emitted
 * (a complex product with (env, 0)); a MUTED item is dropped when gating, else emitted as zero.  Returns the output count. */
size_t orc_pwr_squelch_cc(const cf32* in, size_t n, double db, double alpha, int ramp, int gate, cf32* out)
{
    orc_trace_event("pwr_squelch_cc(%.17g,%.17g,%d,%d)", db, alpha, ramp, gate);
    const double threshold = pow(10.0, db / 10);
    double pwr = 0.0;
    int state = 0, ramped = 0;
    float env = ramp ? 0.0f : 1.0f;
    float* tab = NEW(float, ramp + 1);
    if (ramp) orc_squelch_envelope(ramp, tab);
    size_t j = 0;
    for (size_t i = 0; i < n; i++) {
        const float p = in[i].re * in[i].re + in[i].im * in[i].im;
        pwr = alpha * (double)p + (1.0 - alpha) * pwr;
        const int mute = pwr < threshold;
        switch (state) {
        case 0: if (!mute) state = ramp ? 1 : 2; break;
        case 2: if (mute) state = ramp ? 3 : 0; break;
        case 1:
            env = tab[++ramped];
            if (ramped >= ramp) { state = 2; env = 1.0f; }
            break;
        case 3:
            env = tab[--ramped];
            if (ramped == 0) state = 0;
            break;
        }
        if (state != 0) {
            out[j].re = in[i].re * env - in[i].im * 0.0f;
            out[j].im = in[i].re * 0.0f + in[i].im * env;
            j++;
        } else if (!gate) {
            out[j].re = 0.0f; out[j].im = 0.0f; j++;
        }
    }
    free(tab);
    return j;
}

/* ------------------------------------------------------------------------------------------
 * analog::ctcss_squelch_ff(rate, freq, level, len, ramp, gate)  [GR-MEM: gr-analog/lib/ctcss_squelch_ff_impl.cc, squelch_base_ff_impl.cc,
 * gr-fft/lib/goertzel.cc] -- instance make(8000, 88.5, 0.01, 8000, 160, true) src/gr/gr_demod_nbfm.cpp:59-60, switched into the
 * audio path by gr_demod_nbfm::set_ctcss(tone) (:97-123).
 *   three Goertzel filters over blocks of `len` items: the tone and its neighbours in the CTCSS table (a tone not in the table or at
 *   its ends: -2 % / +2 %); per item  y = x + wr d1 - d2 (float, left to right), d2 = d1, d1 = y,  w = (float)(2 pi f / rate),
 *   wr = (float)(2.0 * cosf(w)), wi = sinf(w); after `len` items  out = ((0.5 wr d1 - d2) / len [in double, then float], (wi d1) / len),
 *   |out| as sqrtf(re^2 + im^2) (std::abs of a complex<float>; hypot's last bit is libm's business), floorf(1e5 |out|) / 1e5 per
 *   filter, mute = c < level || c < l || c < r (level compared as double); the filters restart.
 *   squelch_base_ff: MUTED / ATTACK / UNMUTED / DECAY with the raised-cosine envelope 0.5 - cos(pi k / ramp) / 2 (double);
 *   an unmuted item leaves as (float)((double)x * envelope), a muted one is dropped (gate) or leaves as 0.
 * st (carried between calls): [0..5] d1, d2 of the l, c, r filters, [6] processed, [7] mute, [8] state, [9] ramped, env in *env.
 * ------------------------------------------------------------------------------------------ */
static const float ctcss_tones[38] = {67.0f, 71.9f, 74.4f, 77.0f, 79.7f, 82.5f, 85.4f, 88.5f, 91.5f, 94.8f, 97.4f, 100.0f, 103.5f, 107.2f, 110.9f, 114.8f,
                                      118.8f, 123.0f, 127.3f, 131.8f, 136.5f, 141.3f, 146.2f, 151.4f, 156.7f, 162.2f, 167.9f, 173.8f, 179.9f, 186.2f, 192.8f,
                                      203.5f, 210.7f, 218.1f, 225.7f, 233.6f, 241.8f, 250.3f};
void orc_ctcss_freqs(float freq, float* f_l, float* f_r)
{
    int i = -1;
    for (int k = 0; k < 38; k++) if (ctcss_tones[k] == freq) i = k;
    /* the guard tones of an off-table tone: freq * 0.98 / freq * 1.02 with DOUBLE literals, narrowed to float afterwards [GR-MEM] (the reference's
     * tone list has off-table entries: 81.5 and 87.4, src/ext/utils.h:17) */
    *f_l = (i == -1 || i == 0) ? (float)((double)freq * 0.98) : ctcss_tones[i - 1];
    *f_r = (i == -1 || i == 37) ? (float)((double)freq * 1.02) : ctcss_tones[i + 1];
}
void orc_goertzel_coeffs(int rate, float freq, float* wr, float* wi)
{
    const float w = (float)(2.0 * M_PI * freq / rate);
    *wr = (float)(2.0 * (double)cosf(w));
    *wi = sinf(w);
}
size_t orc_ctcss_squelch_ff(const float* in, size_t n, int rate, float freq, double level, int len, int ramp, int gate, float* out)
{
    orc_trace_event("ctcss_squelch_ff(%d,%.9g,%.17g,%d,%d,%d)", rate, freq, level, len, ramp, gate);
    if (len == 0) len = (int)(rate / 10.0);
    float fl, fr, wr[3], wi[3];
    orc_ctcss_freqs(freq, &fl, &fr);
    orc_goertzel_coeffs(rate, fl, &wr[0], &wi[0]);
    orc_goertzel_coeffs(rate, freq, &wr[1], &wi[1]);
    orc_goertzel_coeffs(rate, fr, &wr[2], &wi[2]);
    float d1[3] = {0, 0, 0}, d2[3] = {0, 0, 0};
    int processed = 0, mute = 1, state = 0, ramped = 0;
    double env = ramp ? 0.0 : 1.0;
    size_t j = 0;
    for (size_t i = 0; i < n; i++) {
        for (int k = 0; k < 3; k++) {
            float y = in[i] + wr[k] * d1[k];
            y = y - d2[k];
            d2[k] = d1[k]; d1[k] = y;
        }
        if (++processed == len) {
            float o[3];
            for (int k = 0; k < 3; k++) {
                const float re = (float)((0.5 * (double)wr[k] * (double)d1[k] - (double)d2[k]) / (double)len);
                const float im = (wi[k] * d1[k]) / (float)len;
                o[k] = floorf(100000.0f * sqrtf(re * re + im * im)) / 100000.0f;
                d1[k] = d2[k] = 0.0f;
            }
            processed = 0;
            mute = ((double)o[1] < level) || o[1] < o[0] || o[1] < o[2];
        }
        switch (state) {
        case 0: if (!mute) state = ramp ? 1 : 2; break;
        case 2: if (mute) state = ramp ? 3 : 0; break;
        case 1:
            env = 0.5 - cos(M_PI * (double)(++ramped) / (double)ramp) / 2.0;
            if (ramped >= ramp) { state = 2; env = 1.0; }
            break;
        case 3:
            env = 0.5 - cos(M_PI * (double)(--ramped) / (double)ramp) / 2.0;
            if (ramped == 0) state = 0;
            break;
        }
        if (state != 0) out[j++] = (float)((double)in[i] * env);
        else if (!gate) out[j++] = 0.0f;
    }
    return j;
}

/* agc2_ff(attack, decay, reference, gain), max gain 65536 */
void orc_agc2_ff(const float* in, size_t n, float attack, float decay, float ref, float gain, float max_gain, float* out)
{
    orc_trace_event("agc2_ff(%.9g,%.9g,%.9g,%.9g,%.9g)", attack, decay, ref, gain, max_gain);
    for (size_t i = 0; i < n; i++) {
        const float o = in[i] * gain;
        const float tmp = -ref + fabsf(o);
        float rate = decay;
        if (fabsf(tmp) > gain) rate = attack;
        gain -= tmp * rate;
        if (gain < 0.0f) gain = 10e-5f;
        if (max_gain > 0.0f && gain > max_gain) gain = max_gain;
        out[i] = o;
    }
}

/* iir_filter_ffd with two feed-forward and two feedback taps: acc = ff0 x + ff1 x[-1] + fb1 y[-1] in double (y kept in double),
 * output (float)acc.  oldstyle: fb taps as given; new style: negated (y = sum b x - sum a y). */
void orc_iir_ffd_2(const float* in, size_t n, const double ff[2], const double fb[2], int oldstyle, float* out)
{
    orc_trace_event("iir_ffd(%.17g,%.17g,%.17g,%.17g,%d)", ff[0], ff[1], fb[0], fb[1], oldstyle);
    const double fb1 = oldstyle ? fb[1] : -fb[1];
    float xp = 0.0f; double yp = 0.0;
    for (size_t i = 0; i < n; i++) {
        double acc = ff[0] * (double)in[i];
        acc += ff[1] * (double)xp;
        acc += fb1 * yp;
        yp = acc; xp = in[i];
        out[i] = (float)acc;
    }
}

static void scale_f(float* x, size_t n, float k) { for (size_t i = 0; i < n; i++) x[i] = x[i] * k; }

/* gr_demod_nbfm::set_ctcss(value) for the NEXT orc_demod_analog(kind 0) calls: 0 = off (the constructor's graph) */
static float g_ctcss_tone = 0.0f;
void orc_set_ctcss(float tone_hz) { g_ctcss_tone = tone_hz; }
/* gr_demod_nbfm / _am / _wbfm / _ssb::set_filter_width(width) (what gr_demod_base::set_filter_width(width, mode) forwards, src/gr/gr_demod_base.cpp:1155-1185) for
 * the NEXT orc_demod_analog / orc_demod_ssb calls; 0 = the constructor's graph.  The setters do not repeat the constructors' designs:
 *   gr_demod_nbfm.cpp:82-90   _filter low_pass(1, 20000, w, 1200, BH), _fm_demod gain 20000 / (4 pi w)          (constructor: low_pass_2(.., 3500, 60))
 *   gr_demod_am.cpp:84-91     _filter complex_band_pass(1, 20000, -w, w, 1200, BH)                              (constructor: complex_band_pass_2(.., 200, 90))
 *   gr_demod_wbfm.cpp:76-84   _filter low_pass(1, 200000, w, 1200, BH), _fm_demod gain 200000 / (2 pi w)        (constructor: low_pass_2(.., 600, 90))
 *   gr_demod_ssb.cpp:89-101   both band-pass filters as constructed with w, the audio filter band_pass_2(2, 8000, 200, w, 200, 90, BH): GAIN 2 (constructor: 1)
 * orc_set_rx_gain: gr_demod_ssb::set_gain(value) = _if_gain->set_k(value) (:118-121; gr_demod_base::set_gain, gr_demod_base.cpp:1206-1210); < 0 = the constructor's 0.9 */
static int g_rx_fw_set = 0;
static float g_rx_gain = -1.0f;
void orc_set_rx_filter_width(int width) { g_rx_fw_set = width; }
void orc_set_rx_gain(float k) { g_rx_gain = k; }
/* kind: 0 NBFM, 1 AM, 2 WBFM.  filtered = port 0 (before the squelch), audio = port 1 (8 kHz). */
void orc_demod_analog(const cf32* in, size_t n, int kind, int samp_rate, int filter_width,
                      cf32** filtered, size_t* n_filtered, float** audio, size_t* n_audio)
{
    const int target = kind == 2 ? 200000 : 20000, decim = kind == 2 ? 5 : 50;
    int nt = orc_low_pass(1, samp_rate, target / 2, target / 2, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* taps = NEW(float, nt);
    orc_low_pass(1, samp_rate, target / 2, target / 2, ORC_WIN_BLACKMAN_HARRIS, taps);
    const size_t n1 = orc_decim_count(n, 1, decim);
    cf32* s1 = NEW(cf32, n1);
    orc_decim_auto(in, n, taps, nt, decim, s1);                                     /* _resampler */
    free(taps);
    cf32* f = NEW(cf32, n1);
    const int wset = g_rx_fw_set > 0;
    if (wset) filter_width = g_rx_fw_set;
    if (wset && kind == 1) {
        int nf = orc_complex_band_pass(1, target, -filter_width, filter_width, 1200, ORC_WIN_BLACKMAN_HARRIS, NULL);
        cf32* ft = NEW(cf32, nf);
        orc_complex_band_pass(1, target, -filter_width, filter_width, 1200, ORC_WIN_BLACKMAN_HARRIS, ft);
        orc_fir_ccc(s1, n1, ft, nf, f);
        free(ft);
    } else if (wset) {
        int nf = orc_low_pass(1, target, filter_width, 1200, ORC_WIN_BLACKMAN_HARRIS, NULL);
        float* ft = NEW(float, nf);
        orc_low_pass(1, target, filter_width, 1200, ORC_WIN_BLACKMAN_HARRIS, ft);
        orc_fir_ccf(s1, n1, ft, nf, f);
        free(ft);
    } else if (kind == 1) {                                                         /* _filter: fft_filter_ccc */
        int nf = orc_complex_band_pass_2(1, target, -filter_width, filter_width, 200, 90, ORC_WIN_BLACKMAN_HARRIS, NULL);
        cf32* ft = NEW(cf32, nf);
        orc_complex_band_pass_2(1, target, -filter_width, filter_width, 200, 90, ORC_WIN_BLACKMAN_HARRIS, ft);
        orc_fir_ccc(s1, n1, ft, nf, f);
        free(ft);
    } else {                                                                        /* _filter: fft_filter_ccf */
        const double tw = kind == 0 ? 3500 : 600, att = kind == 0 ? 60 : 90;
        int nf = orc_low_pass_2(1, target, filter_width, tw, att, ORC_WIN_BLACKMAN_HARRIS, NULL);
        float* ft = NEW(float, nf);
        orc_low_pass_2(1, target, filter_width, tw, att, ORC_WIN_BLACKMAN_HARRIS, ft);
        orc_fir_ccf(s1, n1, ft, nf, f);
        free(ft);
    }
    free(s1);
    *filtered = f; *n_filtered = n1;
    cf32* g = NEW(cf32, n1);
    const size_t ng = orc_pwr_squelch_cc(f, n1, -140, 0.01, kind == 0 ? 320 : 0, 1, g);   /* _squelch (gating) */
    float* d = NEW(float, ng);
    if (kind == 1) {
        for (size_t i = 0; i < ng; i++) d[i] = sqrtf(g[i].re * g[i].re + g[i].im * g[i].im);   /* _complex_to_mag */
        orc_agc2_ff(d, ng, 1e-1f, 1e-1f, 1.0f, 1.0f, 65536.0f, d);                             /* _agc */
        const double ff[2] = {1, -1}, fb[2] = {0, 0.9999};
        orc_iir_ffd_2(d, ng, ff, fb, 1, d);                                                     /* _iir_filter (DC block) */
        scale_f(d, ng, 0.99f);                                                                  /* _audio_gain */
    } else {
        const float gain = (float)(target / ((kind == 0 ? 4 : 2) * M_PI * filter_width));
        orc_quad_demod(g, ng, gain, d);                                                         /* _fm_demod */
    }
    free(g);
    double a[2], b[2];
    float* out;
    size_t no;
    if (kind == 2) {
        scale_f(d, ng, 0.9f);                                                                   /* _amplify */
        orc_deemph_taps(8000, 50e-6, a, b);
        orc_iir_ffd_2(d, ng, b, a, 0, d);                                                       /* _de_emph_filter (at 200 ksps) */
        int na = orc_low_pass(1, target, 4000, 2000, ORC_WIN_BLACKMAN_HARRIS, NULL);
        float* at = NEW(float, na);
        orc_low_pass(1, target, 4000, 2000, ORC_WIN_BLACKMAN_HARRIS, at);
        no = orc_decim_count(ng, 1, 25);
        out = NEW(float, no);
        orc_resamp_fff(d, ng, at, na, 1, 25, out);                                              /* _audio_resampler (1, 25) */
        free(at);
    } else {
        int na = kind == 0 ? orc_low_pass_2(2, 2 * target, 3600, 250, 60, ORC_WIN_BLACKMAN_HARRIS, NULL)
                           : orc_low_pass(2, 2 * target, 3600, 600, ORC_WIN_BLACKMAN_HARRIS, NULL);
        float* at = NEW(float, na);
        if (kind == 0) orc_low_pass_2(2, 2 * target, 3600, 250, 60, ORC_WIN_BLACKMAN_HARRIS, at);
        else           orc_low_pass(2, 2 * target, 3600, 600, ORC_WIN_BLACKMAN_HARRIS, at);
        no = orc_decim_count(ng, 2, 5);
        float* r = NEW(float, no);
        orc_resamp_fff(d, ng, at, na, 2, 5, r);                                                 /* _audio_resampler (2, 5) */
        free(at);
        const int ctcss = kind == 0 && g_ctcss_tone != 0.0f;      /* gr_demod_nbfm::set_ctcss(tone): _ctcss between resampler and audio filter, band-pass audio filter */
        if (ctcss) no = orc_ctcss_squelch_ff(r, no, 8000, g_ctcss_tone, 0.01, 8000, 160, 1, r);  /* (in place: j <= i) */
        int nf = ctcss ? orc_band_pass_2(1, 8000, 300, 3500, 200, 35, ORC_WIN_BLACKMAN_HARRIS, NULL)
               : kind == 0 ? orc_low_pass_2(1, 8000, 3500, 200, 35, ORC_WIN_BLACKMAN_HARRIS, NULL)
                           : orc_low_pass(1, 8000, 3600, 300, ORC_WIN_BLACKMAN_HARRIS, NULL);
        float* aft = NEW(float, nf);
        if (ctcss) orc_band_pass_2(1, 8000, 300, 3500, 200, 35, ORC_WIN_BLACKMAN_HARRIS, aft);
        else if (kind == 0) orc_low_pass_2(1, 8000, 3500, 200, 35, ORC_WIN_BLACKMAN_HARRIS, aft);
        else           orc_low_pass(1, 8000, 3600, 300, ORC_WIN_BLACKMAN_HARRIS, aft);
        out = NEW(float, no);
        orc_fir_fff(r, no, aft, nf, out);                                                       /* _audio_filter */
        free(aft); free(r);
        if (kind == 0) {
            orc_deemph_taps(target, 50e-6, a, b);
            orc_iir_ffd_2(out, no, b, a, 0, out);                                               /* _de_emph_filter */
            scale_f(out, no, 2.0f);                                                             /* _level_control */
        }
    }
    free(d);
    *audio = out; *n_audio = no;
}

/* cessb::clipper_cc(clip): polar clip, item by item (the block works in chunks of 1024, the values do not depend on them):
 * magnitude sqrtf(re^2 + im^2), phase fast_atan2f, min(magnitude, clip), back through cos / sin.  volk_32f_cos_32f / sin_32f are
 * restated with the oracle's own deterministic sincos (orc_sincosf): [GR-MEM] the VOLK kernels are polynomial approximations whose
 * last bits depend on the machine's SIMD path anyway. */
void orc_cessb_clipper(const cf32* in, size_t n, float clip, cf32* out)
{
    for (size_t i = 0; i < n; i++) {
        const float mag = sqrtf(in[i].re * in[i].re + in[i].im * in[i].im);
        const float ph = orc_fast_atan2f(in[i].im, in[i].re);
        const float c = mag < clip ? mag : clip;
        float sn, cs;
        orc_sincosf(ph, &sn, &cs);
        out[i].re = cs * c; out[i].im = sn * c;
    }
}
/* cessb::stretcher_cc: output k = in[k] / h, h = (max(emax max(|in[k-2 .. k+2]|), 1) - 1) 2 + 1, emax = 1 / (sqrt(0.5) / 2) as float;
 * items before the stream start count as 0.  The block emits whole chunks of 1024 and reads two items ahead: n input items give
 * 1024 floor((n - 2) / 1024) outputs (returned). */
size_t orc_cessb_stretcher(const cf32* in, size_t n, cf32* out)
{
    const float emax = (float)(1 / (sqrt(0.5) / 2));
    const size_t nout = n >= 2 ? 1024 * ((n - 2) / 1024) : 0;
    for (size_t k = 0; k < nout; k++) {
        float e = 0.0f;
        for (long long j = (long long)k - 2; j <= (long long)k + 2; j++) {
            if (j < 0) continue;
            const float m = sqrtf(in[j].re * in[j].re + in[j].im * in[j].im);
            e = m > e ? m : e;
        }
        float h = e * emax;
        h = h > 1.0f ? h : 1.0f;
        h = h - 1.0f;
        h = h * 2.0f;
        h = h + 1.0f;
        out[k].re = in[k].re / h; out[k].im = in[k].im / h;
    }
    return nout;
}

/* gr_demod_ssb(sps = 125, ., ., filter_width, sb): 1:125 to 8 ksps -> x0.9 -> complex band-pass (port 0) -> gating squelch ->
 * agc2_cc(0.1, 0.1, 0.25, 1) -> clipper(0.95) -> stretcher -> real part -> x1.333 -> audio band-pass (port 1) */
void orc_demod_ssb(const cf32* in, size_t n, int samp_rate, int filter_width, int sb,
                   cf32** filtered, size_t* n_filtered, float** audio, size_t* n_audio)
{
    const int target = 8000, decim = 125;
    int nt = orc_low_pass(1, samp_rate, target / 2, target / 2, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* taps = NEW(float, nt);
    orc_low_pass(1, samp_rate, target / 2, target / 2, ORC_WIN_BLACKMAN_HARRIS, taps);
    const size_t n1 = orc_decim_count(n, 1, decim);
    cf32* s1 = NEW(cf32, n1);
    orc_decim_auto(in, n, taps, nt, decim, s1);                                                 /* _resampler */
    free(taps);
    const float ifg = g_rx_gain >= 0.0f ? g_rx_gain : 0.9f;
    for (size_t i = 0; i < n1; i++) { s1[i].re = s1[i].re * ifg; s1[i].im = s1[i].im * ifg; }   /* _if_gain */
    const double ag = g_rx_fw_set > 0 ? 2 : 1;                                                  /* set_filter_width's audio filter has gain 2 */
    if (g_rx_fw_set > 0) filter_width = g_rx_fw_set;
    const double lo = sb ? -filter_width : 200, hi = sb ? -200 : filter_width;
    int nf = orc_complex_band_pass_2(1, target, lo, hi, 200, 90, ORC_WIN_BLACKMAN_HARRIS, NULL);
    cf32* ft = NEW(cf32, nf);
    orc_complex_band_pass_2(1, target, lo, hi, 200, 90, ORC_WIN_BLACKMAN_HARRIS, ft);
    cf32* f = NEW(cf32, n1);
    orc_fir_ccc(s1, n1, ft, nf, f);                                                             /* _filter_usb / _filter_lsb -> port 0 */
    free(ft); free(s1);
    *filtered = f; *n_filtered = n1;
    cf32* g = NEW(cf32, n1);
    const size_t ng = orc_pwr_squelch_cc(f, n1, -140, 0.01, 0, 1, g);                           /* _squelch */
    cf32* a = NEW(cf32, ng);
    orc_agc2(g, ng, 1e-1f, 1e-1f, 0.25f, 1.0f, 65536.0f, a);                                    /* _agc */
    orc_cessb_clipper(a, ng, 0.95f, g);                                                         /* _clipper */
    const size_t ns = orc_cessb_stretcher(g, ng, a);                                            /* _stretcher */
    float* r = NEW(float, ns);
    for (size_t i = 0; i < ns; i++) r[i] = a[i].re * 1.333f;                                    /* _complex_to_real, _level_control */
    free(g); free(a);
    int na = orc_band_pass_2(ag, target, 200, filter_width, 200, 90, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* at = NEW(float, na);
    orc_band_pass_2(ag, target, 200, filter_width, 200, 90, ORC_WIN_BLACKMAN_HARRIS, at);
    float* out = NEW(float, ns);
    orc_fir_fff(r, ns, at, na, out);                                                            /* _audio_filter -> port 1 */
    free(at); free(r);
    *audio = out; *n_audio = ns;
}

/* gr_mod_ssb(sps = 125, 1000000, ., filter_width, sb) (reference src/gr/gr_mod_ssb.cpp:26-82, instances gr_mod_base.cpp:178-179):
 * audio at 8 ksps -> fft_filter_fff(band_pass_2(1, 8000, 300, fw, 200, 90, BH)) -> float_to_complex -> cessb clipper(0.95) -> stretcher
 * -> fft_filter_ccc(complex_band_pass_2(1, 8000, 200, fw | -fw, -200, 200, 90, BH)) -> x0.9 -> x bb_gain ->
 * rational_resampler_ccf(125, 1, low_pass_2(125, 1e6, fw, fw, 90, BH)).  The stretcher emits whole chunks of 1024 and reads two
 * items ahead: n audio samples give 125 * 1024 floor((n - 2) / 1024) IQ samples. */
size_t orc_mod_ssb(const float* audio, size_t n, int sps, int samp_rate, int filter_width, int sb, float bb_gain, cf32* out)
{
    const size_t ns = n >= 2 ? 1024 * ((n - 2) / 1024) : 0;
    if (!out) return ns * (size_t)sps;
    int na = orc_band_pass_2(1, 8000, 300, filter_width, 200, 90, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* at = NEW(float, na);
    orc_band_pass_2(1, 8000, 300, filter_width, 200, 90, ORC_WIN_BLACKMAN_HARRIS, at);
    float* a1 = NEW(float, n);
    orc_fir_fff(audio, n, at, na, a1);                                       /* _audio_filter */
    free(at);
    cf32* c = NEW(cf32, n);
    for (size_t i = 0; i < n; i++) { c[i].re = a1[i]; c[i].im = 0.0f; }      /* _float_to_complex */
    free(a1);
    cf32* d = NEW(cf32, n);
    orc_cessb_clipper(c, n, 0.95f, d);                                       /* _clipper */
    orc_cessb_stretcher(d, n, c);                                            /* _stretcher: ns items */
    extern int g_tx_fw_set;                                                  /* orc_set_tx_filter_width (orc_chains.c): the audio filter above keeps the constructor's width */
    const int wset = g_tx_fw_set > 0;
    if (wset) filter_width = g_tx_fw_set;
    const double e0 = wset ? 300 : 200, tw = wset ? 250 : 200;
    const double lo = sb ? -filter_width : e0, hi = sb ? -e0 : filter_width;
    int nf = orc_complex_band_pass_2(1, 8000, lo, hi, tw, 90, ORC_WIN_BLACKMAN_HARRIS, NULL);
    cf32* ft = NEW(cf32, nf);
    orc_complex_band_pass_2(1, 8000, lo, hi, tw, 90, ORC_WIN_BLACKMAN_HARRIS, ft);
    orc_fir_ccc(c, ns, ft, nf, d);                                           /* _filter_usb / _filter_lsb */
    free(ft); free(c);
    for (size_t i = 0; i < ns; i++) { d[i].re *= 0.9f; d[i].im *= 0.9f; d[i].re *= bb_gain; d[i].im *= bb_gain; }   /* _amplify, _bb_gain */
    int nt = orc_low_pass_2(sps, samp_rate, filter_width, filter_width, 90, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* lp = NEW(float, nt);
    orc_low_pass_2(sps, samp_rate, filter_width, filter_width, 90, ORC_WIN_BLACKMAN_HARRIS, lp);
    const size_t m = orc_resamp_ccf(d, ns, lp, nt, sps, 1, out);             /* _resampler */
    free(lp); free(d);
    return m;
}

void orc_free(void* p) { free(p); }


/* ------------------------------------------------------------------------------------------
 * gr_mod_am (reference src/gr/gr_mod_am.cpp:26-74, instance make_gr_mod_am(125, 1000000, 1700, 5000) src/gr/gr_mod_base.cpp:167):
 *   audio (8 ksps) -> agc2_ff(1e-2, 1e-4, 1, 1), max gain 1 -> rail_ff(-0.98, 0.98) -> x0.95 -> fft_filter_fff(band_pass_2(1, 8000,
 *   300, 3000, 200, 60, HAMMING)) -> + carrier -> float_to_complex -> rational_resampler_ccf(sps, 1, low_pass(sps, samp_rate, fw, fw))
 *   -> x0.5 -> x bb_gain -> fft_filter_ccc(complex_band_pass_2(1, samp_rate, -fw, fw, 1200, 120, BH)).
 * The carrier is sig_source_f(8000, GR_COS_WAVE, 0, 0.5): frequency 0, so a constant 0.5 cos(0).  [GR-MEM] upstream's sig_source_f
 * takes the cosine from gr::fxpt's interpolated sine table; cos(0) is taken as exactly 1 here (the table's value at the quarter
 * turn may differ from 1 in the last bit: 6e-8 relative, inside the 1e-5 bound on float samples).  _feed_forward_agc is created by
 * the constructor but never connected (:66-77): it is not part of the graph.
 * ------------------------------------------------------------------------------------------ */
size_t orc_mod_am(const float* audio, size_t n, int sps, int samp_rate, int filter_width, float bb_gain, cf32* out)
{
    if (!out) return n * (size_t)sps;
    float* a0 = NEW(float, n ? n : 1);
    orc_agc2_ff(audio, n, 1e-2f, 1e-4f, 1.0f, 1.0f, 1.0f, a0);                    /* _agc, set_max_gain(1.0) */
    for (size_t i = 0; i < n; i++) {                                              /* _rail, _audio_amplify */
        float v = a0[i];
        if (v < -0.98f) v = -0.98f; else if (v > 0.98f) v = 0.98f;
        a0[i] = v * 0.95f;
    }
    int na = orc_band_pass_2(1, 8000, 300, 3000, 200, 60, ORC_WIN_HAMMING, NULL);
    float* at = NEW(float, na);
    orc_band_pass_2(1, 8000, 300, 3000, 200, 60, ORC_WIN_HAMMING, at);
    float* a1 = NEW(float, n ? n : 1);
    orc_fir_fff(a0, n, at, na, a1);                                               /* _audio_filter */
    free(at); free(a0);
    cf32* c = NEW(cf32, n ? n : 1);
    for (size_t i = 0; i < n; i++) { c[i].re = a1[i] + 0.5f; c[i].im = 0.0f; }    /* _add (+ _signal_source), _float_to_complex */
    free(a1);
    { extern int g_tx_fw_set; if (g_tx_fw_set > 0) filter_width = g_tx_fw_set; }   /* orc_set_tx_filter_width: the constructor's designs with the new width */
    int ni = orc_low_pass(sps, samp_rate, filter_width, filter_width, ORC_WIN_HAMMING, NULL);
    float* it = NEW(float, ni);
    orc_low_pass(sps, samp_rate, filter_width, filter_width, ORC_WIN_HAMMING, it);
    cf32* r = NEW(cf32, n * (size_t)sps + 1);
    const size_t m = orc_resamp_ccf(c, n, it, ni, sps, 1, r);                     /* _resampler */
    free(it); free(c);
    for (size_t i = 0; i < m; i++) { r[i].re *= 0.5f; r[i].im *= 0.5f; r[i].re *= bb_gain; r[i].im *= bb_gain; }   /* _amplify, _bb_gain */
    int nf = orc_complex_band_pass_2(1, samp_rate, -filter_width, filter_width, 1200, 120, ORC_WIN_BLACKMAN_HARRIS, NULL);
    cf32* ft = NEW(cf32, nf);
    orc_complex_band_pass_2(1, samp_rate, -filter_width, filter_width, 1200, 120, ORC_WIN_BLACKMAN_HARRIS, ft);
    orc_fir_ccc(r, m, ft, nf, out);                                               /* _filter */
    free(ft); free(r);
    return m;
}
