/* orc_chains.c — the reference's hier-block topologies wired from oracle blocks
 * (TEST INFRASTRUCTURE; see orc.h).  Each chain cites the reference file:line it follows. */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NEW(T, n) ((T*)calloc((size_t)(n) + 16, sizeof(T)))

void orc_demod_out_free(orc_demod_out* o)
{
    free(o->filtered); free(o->constellation); free(o->bits_a); free(o->bits_b);
    memset(o, 0, sizeof *o);
}

/* ---- RX front end: gr_demod_base.cpp:57,180 (rotator), :1220-1225 (phase inc = 2*pi*-off/fs),
 *      :1330-1340 (rational_resampler_ccf(1, fs/1e6, low_pass(1, fs, 480k, 100k, BH)) when fs >= 2 Msps) */
int orc_frontend_taps(int samp_rate, float* taps)
{
    if (samp_rate < 2000000) return 0;
    return orc_low_pass(1, samp_rate, 480000, 100000, ORC_WIN_BLACKMAN_HARRIS, taps);
}
size_t orc_frontend(const cf32* in, size_t n, int samp_rate, double carrier_offset_hz, cf32* out)
{
    uint64_t inc = orc_phase_inc_to_turn(2 * M_PI * -carrier_offset_hz / samp_rate);
    if (samp_rate < 2000000) { orc_rotator(in, n, inc, 0, out); return n; }
    int decim = samp_rate / 1000000;
    int nt = orc_frontend_taps(samp_rate, NULL);
    float* taps = NEW(float, nt);
    orc_frontend_taps(samp_rate, taps);
    cf32* rot = NEW(cf32, n);
    orc_rotator(in, n, inc, 0, rot);
    size_t m = orc_decim_auto(rot, n, taps, nt, decim, out);
    free(rot); free(taps);
    return m;
}

/* soft symbols -> {viterbi -> descrambler} on the stream and on the stream delayed by one
 * (gr_demod_2fsk.cpp:155-164, gr_demod_gmsk.cpp:122-131) */
static void fec_tail(const float* sym, size_t nsym, float mul, int two_branch, orc_demod_out* o)
{
    uint8_t* soft = NEW(uint8_t, nsym + 1);
    orc_soft_quant(sym, nsym, mul, 128.0f, soft + 1);
    soft[0] = 0; /* blocks::delay(1): one zero item in front */
    uint8_t* dec = NEW(uint8_t, nsym / 2 + 80);
    size_t nb = orc_cc_decode_k7(soft + 1, nsym, dec);
    o->bits_a = NEW(uint8_t, nb); o->n_bits_a = nb;
    orc_descramble(dec, nb, 0x8A, 0x7F, 7, o->bits_a);
    if (two_branch) {
        nb = orc_cc_decode_k7(soft, nsym + 1, dec);
        o->bits_b = NEW(uint8_t, nb); o->n_bits_b = nb;
        orc_descramble(dec, nb, 0x8A, 0x7F, 7, o->bits_b);
    }
    free(soft); free(dec);
}

/* gr_demod_2fsk.cpp:38-167 */
void orc_demod_2fsk(const cf32* in, size_t n, int sps, int samp_rate, int carrier_freq, int filter_width, int fm, orc_demod_out* o)
{
    (void)carrier_freq;
    memset(o, 0, sizeof *o);
    int decim, interp, nfilts, target, sps_eff;
    if (sps == 10)      { target = 20000; sps_eff = sps;     decim = 50; interp = 1; nfilts = 35 * sps_eff; }
    else if (sps >= 5)  { target = 40000; sps_eff = sps * 2; decim = 25; interp = 1; nfilts = 35 * sps_eff; }
    else                { target = 80000; sps_eff = 4;       decim = 25; interp = 2; nfilts = 125 * sps_eff; }
    int spacing = fm ? 1 : 2;
    if ((nfilts % 2) == 0) nfilts += 1;

    int nt = orc_low_pass(interp, (double)interp * samp_rate, target / 2, target / 2, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* taps = NEW(float, nt);
    orc_low_pass(interp, (double)interp * samp_rate, target / 2, target / 2, ORC_WIN_BLACKMAN_HARRIS, taps);
    size_t n1 = orc_decim_count(n, interp, decim);
    cf32* s1 = NEW(cf32, n1);
    if (interp == 1) orc_decim_auto(in, n, taps, nt, decim, s1);
    else             orc_resamp_ccf(in, n, taps, nt, interp, decim, s1);
    free(taps);

    cf32* s2 = NEW(cf32, n1);
    orc_fll_band_edge(s1, n1, (float)sps_eff, 0.1f, 16, (float)(24 * M_PI / 100), s2);
    free(s1);

    int nf = orc_low_pass(1, target, filter_width, filter_width, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* ft = NEW(float, nf);
    orc_low_pass(1, target, filter_width, filter_width, ORC_WIN_BLACKMAN_HARRIS, ft);
    o->filtered = NEW(cf32, n1); o->n_filtered = n1;
    orc_fir_ccf(s2, n1, ft, nf, o->filtered);
    free(ft); free(s2);

    float* s5 = NEW(float, n1);
    if (fm) {
        float* dem = NEW(float, n1);
        orc_quad_demod(o->filtered, n1, (float)(sps_eff / (spacing * M_PI / 2)), dem);
        int nr = orc_root_raised_cosine(1, target, target / sps_eff, 0.2, nfilts, NULL);
        float* rrc = NEW(float, nr);
        orc_root_raised_cosine(1, target, target / sps_eff, 0.2, nfilts, rrc);
        orc_fir_fff(dem, n1, rrc, nr, s5);
        free(rrc); free(dem);
    } else {
        int nb = orc_complex_band_pass(1, target, -filter_width, 0, filter_width, ORC_WIN_BLACKMAN_HARRIS, NULL);
        cf32* up = NEW(cf32, nb); cf32* lo = NEW(cf32, nb);
        orc_complex_band_pass(1, target, -filter_width, 0, filter_width, ORC_WIN_BLACKMAN_HARRIS, up);
        orc_complex_band_pass(1, target, 0, filter_width, filter_width, ORC_WIN_BLACKMAN_HARRIS, lo);
        cf32* fu = NEW(cf32, n1); cf32* fl = NEW(cf32, n1);
        orc_fir_ccc_conj_pair(o->filtered, n1, up, lo, nb, fu, fl);
        float* s4 = NEW(float, n1);
        for (size_t i = 0; i < n1; i++) {
            float mu = sqrtf(fu[i].re * fu[i].re + fu[i].im * fu[i].im);
            float ml = sqrtf(fl[i].re * fl[i].re + fl[i].im * fl[i].im);
            float d = mu / ml;
            /* rail_ff(0,2); NaN (0/0 on exact-zero input) -> 0, see DESIGN.md */
            float r = d;
            if (!(r >= 0.0f)) r = 0.0f;
            if (r > 2.0f) r = 2.0f;
            s4[i] = r + (-1.0f);
        }
        int ns = orc_low_pass(1.0, target, target / sps_eff, target / sps_eff, ORC_WIN_HAMMING, NULL);
        float* st = NEW(float, ns);
        orc_low_pass(1.0, target, target / sps_eff, target / sps_eff, ORC_WIN_HAMMING, st);
        orc_fir_fff(s4, n1, st, ns, s5);
        free(st); free(s4); free(fu); free(fl); free(up); free(lo);
    }
    float symbol_rate = (float)target / (float)sps_eff;
    float sps_dev = 200.0f / symbol_rate;
    float* sym = NEW(float, n1 / (size_t)(sps_eff > 1 ? sps_eff - 1 : 1) + 16);
    size_t nsym = orc_symbol_sync_ff(s5, n1, ORC_TED_MOD_MM, (float)sps_eff, (float)(2 * M_PI / (symbol_rate / 10)),
                                     1.0f, 0.2869f, sps_dev, ORC_CONST_BPSK, sym);
    free(s5);
    o->constellation = NEW(cf32, nsym); o->n_const = nsym;
    for (size_t i = 0; i < nsym; i++) { o->constellation[i].re = sym[i]; o->constellation[i].im = 0; }
    fec_tail(sym, nsym, 128.0f, 1, o);
    free(sym);
}

/* gr_demod_gmsk.cpp:38-135 */
void orc_demod_gmsk(const cf32* in, size_t n, int sps, int samp_rate, int carrier_freq, int filter_width, orc_demod_out* o)
{
    (void)carrier_freq;
    memset(o, 0, sizeof *o);
    int decim, interp, target, sps_eff;
    if (sps == 10)     { target = 20000; sps_eff = sps;     decim = 50; interp = 1; }
    else if (sps == 5) { target = 40000; sps_eff = sps * 2; decim = 25; interp = 1; }
    else               { target = 80000; sps_eff = 4;       decim = 25; interp = 2; }

    int nt = orc_low_pass(interp, (double)interp * samp_rate, target / 2, target / 2, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* taps = NEW(float, nt);
    orc_low_pass(interp, (double)interp * samp_rate, target / 2, target / 2, ORC_WIN_BLACKMAN_HARRIS, taps);
    size_t n1 = orc_decim_count(n, interp, decim);
    cf32* s1 = NEW(cf32, n1);
    if (interp == 1) orc_decim_auto(in, n, taps, nt, decim, s1);
    else             orc_resamp_ccf(in, n, taps, nt, interp, decim, s1);
    free(taps);

    int nf = orc_low_pass(1, target, filter_width, filter_width, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* ft = NEW(float, nf);
    orc_low_pass(1, target, filter_width, filter_width, ORC_WIN_BLACKMAN_HARRIS, ft);
    o->filtered = NEW(cf32, n1); o->n_filtered = n1;
    orc_fir_ccf(s1, n1, ft, nf, o->filtered);
    free(ft); free(s1);

    float* dem = NEW(float, n1);
    orc_quad_demod(o->filtered, n1, (float)(sps_eff / (M_PI / 2)), dem);
    int ns = orc_low_pass(1, target, target / sps_eff, target / sps_eff, ORC_WIN_HAMMING, NULL);
    float* st = NEW(float, ns);
    orc_low_pass(1, target, target / sps_eff, target / sps_eff, ORC_WIN_HAMMING, st);
    float* s5 = NEW(float, n1);
    orc_fir_fff(dem, n1, st, ns, s5);
    free(st); free(dem);

    float* sym = NEW(float, n1 / (size_t)(sps_eff - 1) + 16);
    size_t nsym = orc_symbol_sync_ff(s5, n1, ORC_TED_MOD_MM, (float)sps_eff, (float)(2 * M_PI / 200.0f),
                                     1.0f, 0.2869f, 0.05f, ORC_CONST_BPSK, sym);
    free(s5);
    o->constellation = NEW(cf32, nsym); o->n_const = nsym;
    for (size_t i = 0; i < nsym; i++) { o->constellation[i].re = sym[i]; o->constellation[i].im = 0; }
    fec_tail(sym, nsym, 128.0f, 1, o);
    free(sym);
}

/* gr_demod_qpsk.cpp:38-157 */
void orc_demod_qpsk(const cf32* in, size_t n, int sps, int samp_rate, int carrier_freq, int filter_width, orc_demod_out* o)
{
    (void)carrier_freq; (void)filter_width;
    memset(o, 0, sizeof *o);
    int decim, interp = 1, target, sps_eff;
    float costas_bw = (float)(M_PI / 200);
    int fll_bw = 2;
    if (sps > 4 && sps < 125) { decim = 25;  sps_eff = sps * 4 / 25; target = 40000; }
    else if (sps >= 125)      { decim = 100; sps_eff = sps / 25;     target = 10000; }
    else                      { decim = 2;   sps_eff = sps;          target = 500000; costas_bw = (float)(M_PI / 400); }

    int nt = orc_low_pass_2(interp, (double)samp_rate * interp, target / 2, target / 10, 60, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* taps = NEW(float, nt);
    orc_low_pass_2(interp, (double)samp_rate * interp, target / 2, target / 10, 60, ORC_WIN_BLACKMAN_HARRIS, taps);
    size_t n1 = orc_decim_count(n, interp, decim);
    cf32* s1 = NEW(cf32, n1);
    orc_decim_auto(in, n, taps, nt, decim, s1);
    free(taps);
    if (sps > 4) {
        cf32* s1b = NEW(cf32, n1);
        orc_fll_band_edge(s1, n1, (float)sps_eff, 0.35f, 32, (float)(fll_bw * M_PI / 100), s1b);
        free(s1); s1 = s1b;
    }
    int nr = orc_root_raised_cosine(sps_eff, sps_eff, 1, 0.35, 11 * sps_eff, NULL);
    float* rrc = NEW(float, nr);
    orc_root_raised_cosine(sps_eff, sps_eff, 1, 0.35, 11 * sps_eff, rrc);
    o->filtered = NEW(cf32, n1); o->n_filtered = n1;
    orc_fir_ccf(s1, n1, rrc, nr, o->filtered);
    free(rrc); free(s1);

    cf32* a = NEW(cf32, n1);
    orc_agc2(o->filtered, n1, 1.0f, 1e-1f, 1.0f, 1.0f, 65536.0f, a);
    cf32* b = NEW(cf32, n1);
    orc_costas(a, n1, (float)(M_PI / 200 / sps_eff), 4, 1, b);
    free(a);
    float symbol_rate = (float)target / (float)sps_eff;
    float sps_dev = 200.0f / symbol_rate;
    cf32* sy = NEW(cf32, n1 / (size_t)(sps_eff > 1 ? sps_eff - 1 : 1) + 16);
    size_t nsym = orc_symbol_sync_cc(b, n1, ORC_TED_MOD_MM, (float)sps_eff, (float)(2 * M_PI / (symbol_rate / 10)),
                                     1.0f, 0.2869f, sps_dev, ORC_CONST_DQPSK, sy);
    free(b);
    cf32* c2 = NEW(cf32, nsym);
    orc_costas(sy, nsym, costas_bw, 4, 1, c2);
    free(sy);
    cf32* dp = NEW(cf32, nsym);
    orc_diff_phasor(c2, nsym, dp);
    free(c2);
    float ang = (float)(-3 * M_PI / 4);
    cf32 rot; rot.re = (float)cos((double)ang); rot.im = (float)sin((double)ang);
    o->constellation = NEW(cf32, nsym); o->n_const = nsym;
    float* il = NEW(float, 2 * nsym);
    for (size_t i = 0; i < nsym; i++) {
        cf32 v;
        v.re = dp[i].re * rot.re - dp[i].im * rot.im;
        v.im = dp[i].re * rot.im + dp[i].im * rot.re;
        o->constellation[i] = v;
        il[2 * i] = v.re; il[2 * i + 1] = v.im;
    }
    free(dp);
    fec_tail(il, 2 * nsym, 48.0f, 0, o);
    free(il);
}

/* ------------------------------- modulators (TX) --------------------------------------
 * bytes -> unpack MSB first -> scrambler(0x8A,0x7F,7) -> cc_encoder(K7, {109,79}) -> map ->
 * symbols -> pulse shaping / FM -> gains -> interpolator.  The oracle encodes every input bit
 * (fec::encoder would hold back a partial 80-bit frame). */
static size_t tx_bits(const uint8_t* bytes, size_t nbytes, uint8_t** coded)
{
    size_t nb = nbytes * 8;
    uint8_t* u = NEW(uint8_t, nb);
    for (size_t i = 0; i < nb; i++) u[i] = (bytes[i >> 3] >> (7 - (i & 7))) & 1;
    uint8_t* s = NEW(uint8_t, nb);
    orc_scramble(u, nb, 0x8A, 0x7F, 7, s);
    *coded = NEW(uint8_t, 2 * nb);
    orc_cc_encode_k7(s, nb, *coded);
    free(u); free(s);
    return 2 * nb;
}

/* frequency_modulator_fc [gr-analog frequency_modulator_fc_impl.cc]; sin/cos by orc_sincosf
 * (upstream: gr::fxpt table) */
static void fm_mod(const float* in, size_t n, float k, cf32* out)
{
    orc_trace_event("freq_mod(%.9g)", k);
    const float F_PI = (float)M_PI;
    float phase = 0;
    for (size_t i = 0; i < n; i++) {
        phase = phase + k * in[i];
        phase = fmodf(phase + F_PI, 2.0f * F_PI) - F_PI;
        orc_sincosf(phase, &out[i].im, &out[i].re);
    }
}

/* gr_mod_nbfm (reference src/gr/gr_mod_nbfm.cpp:26-77, instances make_gr_mod_nbfm(20, 1000000, 1700, 2500 / 5000) gr_mod_base.cpp:171-172):
 * audio at 8 ksps -> fft_filter_fff(low_pass_2(1, 8000, 3500, 200, 35, BH)) -> x0.99 -> iir_filter_ffd(pre-emphasis, new style) ->
 * rational_resampler_fff(25, 4, low_pass_2(25, 200000, fw, 3500, 60, BH)) -> frequency_modulator_fc(4 pi fw / 50000) ->
 * fft_filter_ccf(low_pass_2(1, 50000, fw, 3500, 60, BH)) -> x0.8 -> x bb_gain -> rational_resampler_ccf(sps = 20, 1,
 * low_pass_2(20, 1e6, fw, 3500, 60, BH)).  125 samples per audio sample.  (The CTCSS tone source exists but is not connected.) */
void orc_preemph_taps(int sample_rate, double tau, double a[2], double b[2])
{
    /* emphasis.cpp:44-89 (fh = 0.925 fs / 2 when not given; tanf on float arguments) */
    const double fs = (double)sample_rate, fh = 0.925 * fs / 2.0;
    const double w_cl = 1.0 / tau, w_ch = 2.0 * M_PI * fh;
    const double w_cla = 2.0 * fs * (double)tanf((float)(w_cl / (2.0 * fs)));
    const double w_cha = 2.0 * fs * (double)tanf((float)(w_ch / (2.0 * fs)));
    const double kl = -w_cla / (2.0 * fs), kh = -w_cha / (2.0 * fs);
    const double z1 = (1.0 + kl) / (1.0 - kl), p1 = (1.0 + kh) / (1.0 - kh), b0 = (1.0 - kl) / (1.0 - kh);
    const double w_0dB = 2.0 * M_PI * 0.0;
    const double g = fabs(1.0 - p1 * 1.0 * (cos(-w_0dB) + sin(-w_0dB))) / (b0 * fabs(1.0 - z1 * 1.0 * (cos(-w_0dB) + sin(-w_0dB))));
    b[0] = g * b0 * 1.0; b[1] = g * b0 * -z1;
    a[0] = 1.0;          a[1] = -p1;
}
/* analog::sig_source_f(fs, GR_COS_WAVE, freq, ampl) [GR-MEM: gr-analog/lib/sig_source_impl.cc, gnuradio-runtime fxpt_nco.h / fxpt.h /
 * sine_table.h -- restated from memory, the least certain entry of the appendix]: a 32-bit fixed-point phase accumulator,
 *   inc = (int32)(x 2^31 / pi) with x = (float)(2 pi freq / fs) (float arithmetic, truncated),  out[k] = (float)(cos_fx(k inc) * ampl),
 *   cos_fx(p): u = p + 0x40000000; row = u >> 22; table[row][0] * (float)(u >> 1) + table[row][1]   (two float roundings),
 * a 1024-row table of float (slope, intercept) pairs of the piecewise-linear sine over x' = u >> 1 in [0, 2^31): slope the secant's,
 * intercept the mean of the secant's and the one through the segment's midpoint (gen_sine_table.py).  k0 = index of the first sample. */
void orc_fxpt_sine_table(float* tab /* 1024 x 2 */)
{
    for (int i = 0; i < 1024; i++) {
        const double a = (double)i * 2097152.0, b = (double)(i + 1) * 2097152.0, w = M_PI / 1073741824.0;
        const double fa = sin(a * w), fb = sin(b * w), fm = sin((a + b) / 2 * w);
        const double m = (fb - fa) / (b - a);
        const double c = (3 * a + b) * (fa - fb) / (4 * (b - a)) + (fm + fa) / 2;
        tab[2 * i] = (float)m; tab[2 * i + 1] = (float)c;
    }
}
uint32_t orc_fxpt_phase_inc(double fs, double freq)
{
    float x = (float)(2 * M_PI * freq / fs);                                      /* set_freq(float angle_rate) */
    const float PI_F = (float)M_PI;
    const int d = (int)floorf(x / 2 / PI_F + 0.5f);
    x -= d * 2 * PI_F;
    return (uint32_t)(int32_t)(x * 2147483648.0f / PI_F);
}
/* the GR_SIN_WAVE form with an offset (the CW key's source, gr_mod_base.cpp:144: sig_source_f(8000, GR_SIN_WAVE, 600, 0.001, 1)): sin_fx(p) is cos_fx without the
 * quarter turn, the offset is added to the float sample afterwards (sig_source_impl::work: d_nco.sin(out, n, ampl); out[i] += offset).  [GR-MEM] */
void orc_sig_source_sin(double fs, double freq, double ampl, float offset, uint64_t k0, size_t n, float* out)
{
    static float tab[2048]; static int have = 0;
    if (!have) { orc_fxpt_sine_table(tab); have = 1; }
    const uint32_t inc = orc_fxpt_phase_inc(fs, freq);
    for (size_t k = 0; k < n; k++) {
        const uint32_t u = (uint32_t)((k0 + k) * (uint64_t)inc);
        const float v = tab[2 * (u >> 22)] * (float)(u >> 1) + tab[2 * (u >> 22) + 1];
        float x = (float)((double)v * ampl);
        if (offset != 0.0f) x = x + offset;
        out[k] = x;
    }
}
void orc_sig_source_cos(double fs, double freq, double ampl, uint64_t k0, size_t n, float* out)
{
    static float tab[2048]; static int have = 0;
    if (!have) { orc_fxpt_sine_table(tab); have = 1; }
    const uint32_t inc = orc_fxpt_phase_inc(fs, freq);
    for (size_t k = 0; k < n; k++) {
        const uint32_t u = (uint32_t)((k0 + k) * (uint64_t)inc) + 0x40000000u;
        const float v = tab[2 * (u >> 22)] * (float)(u >> 1) + tab[2 * (u >> 22) + 1];
        out[k] = (float)((double)v * ampl);
    }
}
/* gr_mod_nbfm::set_ctcss(value) (src/gr/gr_mod_nbfm.cpp:101-135) for the NEXT orc_mod_nbfm calls: tone > 0: _audio_amplify 0.85, the audio
 * filter a band-pass band_pass_2(1, 8000, 300, 3500, 200, 35, BH), sig_source_f(8000, GR_COS_WAVE, tone, 0.15) added in front of the
 * pre-emphasis; tone < 0: set_ctcss(0) after it had been on -- the low-pass again, but _audio_amplify 0.98 (:106; the constructor's is 0.99);
 * 0: the constructor's graph. */
static float g_tx_ctcss = 0.0f;
void orc_set_tx_ctcss(float tone_hz) { g_tx_ctcss = tone_hz; }
/* gr_mod_nbfm / gr_mod_am / gr_mod_ssb::set_filter_width(width) (what gr_mod_base::set_filter_width(width, mode) forwards, src/gr/gr_mod_base.cpp:878-905) for
 * the NEXT orc_mod_nbfm / orc_mod_am / orc_mod_ssb calls; 0 = the constructor's graph.  The setters do not repeat the constructors' designs:
 *   gr_mod_nbfm.cpp:78-93   _if_resampler low_pass_2(25, 200000, w, w, 60, BH), _filter low_pass_2(1, 50000, w, 1200, 60, BH), _resampler low_pass_2(sps, fs, w, w, 60, BH),
 *                           sensitivity 4 pi w / 50000  (constructor: transition 3500 everywhere)
 *   gr_mod_am.cpp:75-85     the constructor's two designs with w
 *   gr_mod_ssb.cpp:85-100   _resampler as constructed with w, _filter_usb / _lsb complex_band_pass_2(1, 8000, 300, w | -w, -300, 250, 90, BH) (constructor: 200 .. w, 200);
 *                           the AUDIO filter keeps the constructor's width */
int g_tx_fw_set = 0;
void orc_set_tx_filter_width(int width) { g_tx_fw_set = width; }
size_t orc_mod_nbfm(const float* audio, size_t n, int sps, int samp_rate, int filter_width, float bb_gain, cf32* out)
{
    const size_t n50 = orc_decim_count(n, 25, 4);
    if (!out) return n50 * (size_t)sps;
    const int tone_on = g_tx_ctcss > 0.0f;
    const int wset = g_tx_fw_set > 0;
    if (wset) filter_width = g_tx_fw_set;
    const double tw_if = wset ? filter_width : 3500, tw_f = wset ? 1200 : 3500;
    const float k_audio = tone_on ? 0.85f : (g_tx_ctcss < 0.0f ? 0.98f : 0.99f);
    int na = tone_on ? orc_band_pass_2(1, 8000, 300, 3500, 200, 35, ORC_WIN_BLACKMAN_HARRIS, NULL) : orc_low_pass_2(1, 8000, 3500, 200, 35, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* at = NEW(float, na);
    if (tone_on) orc_band_pass_2(1, 8000, 300, 3500, 200, 35, ORC_WIN_BLACKMAN_HARRIS, at);
    else orc_low_pass_2(1, 8000, 3500, 200, 35, ORC_WIN_BLACKMAN_HARRIS, at);
    float* tone = NEW(float, n + 1);
    if (tone_on) orc_sig_source_cos(8000, (double)g_tx_ctcss, 0.15, 0, n, tone);
    float* a1 = NEW(float, n);
    orc_fir_fff(audio, n, at, na, a1);                                           /* _audio_filter */
    free(at);
    double ta[2], tb[2];
    orc_preemph_taps(8000, 50e-6, ta, tb);
    orc_trace_event("iir_ffd(%.17g,%.17g,%.17g,%.17g,0)", tb[0], tb[1], ta[0], ta[1]);   /* the new-style IIR restated inline below */
    {   /* _audio_amplify, _pre_emph_filter: acc = b0 x + b1 x[-1] - a1 y[-1] in double, y kept in double */
        float xp = 0.0f; double yp = 0.0;
        for (size_t i = 0; i < n; i++) {
            float x = a1[i] * k_audio;
            if (tone_on) x = x + tone[i];                                         /* _add (add_ff) */
            double acc = tb[0] * (double)x;
            acc += tb[1] * (double)xp;
            acc += -ta[1] * yp;
            yp = acc; xp = x;
            a1[i] = (float)acc;
        }
    }
    int ni = orc_low_pass_2(25, 50000.0 * 4, filter_width, tw_if, 60, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* it = NEW(float, ni);
    orc_low_pass_2(25, 50000.0 * 4, filter_width, tw_if, 60, ORC_WIN_BLACKMAN_HARRIS, it);
    float* r = NEW(float, n50);
    orc_resamp_fff(a1, n, it, ni, 25, 4, r);                                     /* _if_resampler */
    free(it); free(a1); free(tone);
    cf32* fmv = NEW(cf32, n50);
    fm_mod(r, n50, (float)(4 * M_PI * filter_width / 50000.0f), fmv);            /* _fm_modulator */
    free(r);
    int nf = orc_low_pass_2(1, 50000, filter_width, tw_f, 60, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* ft = NEW(float, nf);
    orc_low_pass_2(1, 50000, filter_width, tw_f, 60, ORC_WIN_BLACKMAN_HARRIS, ft);
    cf32* g = NEW(cf32, n50);
    orc_fir_ccf(fmv, n50, ft, nf, g);                                            /* _filter */
    free(ft); free(fmv);
    for (size_t i = 0; i < n50; i++) { g[i].re *= 0.8f; g[i].im *= 0.8f; g[i].re *= bb_gain; g[i].im *= bb_gain; }   /* _amplify, _bb_gain */
    int nt = orc_low_pass_2(sps, samp_rate, filter_width, tw_if, 60, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* lp = NEW(float, nt);
    orc_low_pass_2(sps, samp_rate, filter_width, tw_if, 60, ORC_WIN_BLACKMAN_HARRIS, lp);
    const size_t m = orc_resamp_ccf(g, n50, lp, nt, sps, 1, out);                /* _resampler */
    free(lp); free(g);
    return m;
}

/* gr_mod_2fsk.cpp:30-99 */
size_t orc_mod_2fsk(const uint8_t* bytes, size_t nbytes, int sps, int samp_rate, int carrier_freq, int filter_width, int fm, cf32* out)
{
    (void)carrier_freq;
    int nfilts = 25 * sps, spacing = fm ? 1 : 2, second_interp = 10;
    float amplif = fm ? 0.9f : 0.8f;
    if (sps == 5) nfilts *= 5;
    if ((nfilts % 2) == 0) nfilts += 1;
    size_t nsym_total = nbytes * 16;
    size_t nout = nsym_total * (size_t)sps * (size_t)second_interp;
    if (!out) return nout;
    uint8_t* coded; size_t nc = tx_bits(bytes, nbytes, &coded);
    float* sym = NEW(float, nc);
    for (size_t i = 0; i < nc; i++) sym[i] = coded[i] ? 1.0f : -1.0f;
    free(coded);
    size_t n1 = nc * (size_t)sps;
    float* shaped = NEW(float, n1);
    if (fm) {
        int nr = orc_root_raised_cosine(sps, sps, 1, 0.2, nfilts, NULL);
        float* rrc = NEW(float, nr);
        orc_root_raised_cosine(sps, sps, 1, 0.2, nfilts, rrc);
        orc_resamp_fff(sym, nc, rrc, nr, sps, 1, shaped);
        free(rrc);
    } else {
        for (size_t i = 0; i < n1; i++) shaped[i] = sym[i / (size_t)sps];
    }
    free(sym);
    cf32* fmv = NEW(cf32, n1);
    fm_mod(shaped, n1, (float)((spacing * M_PI / 2) / sps), fmv);
    free(shaped);
    for (size_t i = 0; i < n1; i++) { fmv[i].re *= amplif; fmv[i].im *= amplif; }
    int nt = orc_low_pass(second_interp, samp_rate, filter_width, filter_width, ORC_WIN_HAMMING, NULL);
    float* lp = NEW(float, nt);
    orc_low_pass(second_interp, samp_rate, filter_width, filter_width, ORC_WIN_HAMMING, lp);
    size_t m = orc_resamp_ccf(fmv, n1, lp, nt, second_interp, 1, out);
    free(lp); free(fmv);
    return m;
}

/* gr_mod_gmsk.cpp:30-95 */
size_t orc_mod_gmsk(const uint8_t* bytes, size_t nbytes, int sps, int samp_rate, int carrier_freq, int filter_width, cf32* out)
{
    (void)carrier_freq;
    int nfilts = 35, second_interp = 5;
    float amplif = 0.9f;
    if (sps == 10) { sps = 50; second_interp = 1; nfilts = 55; }
    if (sps == 50) nfilts = 55;
    if (sps == 100) nfilts = 35;
    if ((nfilts % 2) == 0) nfilts += 1;
    size_t nout = nbytes * 16 * (size_t)sps * (size_t)second_interp;
    if (!out) return nout;
    uint8_t* coded; size_t nc = tx_bits(bytes, nbytes, &coded);
    float* sym = NEW(float, nc);
    for (size_t i = 0; i < nc; i++) sym[i] = coded[i] ? 1.0f : -1.0f;
    free(coded);
    size_t n1 = nc * (size_t)sps;
    float* g = NEW(float, nfilts);
    orc_gaussian(sps, sps, 0.3, nfilts, g);
    float* shaped = NEW(float, n1);
    orc_resamp_fff(sym, nc, g, nfilts, sps, 1, shaped);
    free(g); free(sym);
    cf32* fmv = NEW(cf32, n1);
    fm_mod(shaped, n1, (float)((M_PI / 2) / sps), fmv);
    free(shaped);
    for (size_t i = 0; i < n1; i++) { fmv[i].re *= amplif; fmv[i].im *= amplif; }
    int nt = orc_low_pass(second_interp, samp_rate, filter_width, filter_width, ORC_WIN_HAMMING, NULL);
    float* lp = NEW(float, nt);
    orc_low_pass(second_interp, samp_rate, filter_width, filter_width, ORC_WIN_HAMMING, lp);
    size_t m = orc_resamp_ccf(fmv, n1, lp, nt, second_interp, 1, out);
    free(lp); free(fmv);
    return m;
}

/* gr_mod_qpsk.cpp:28-89 */
size_t orc_mod_qpsk(const uint8_t* bytes, size_t nbytes, int sps, int samp_rate, int carrier_freq, int filter_width, cf32* out)
{
    (void)carrier_freq; (void)samp_rate; (void)filter_width;
    int nfilts = (sps > 120) ? 11 : (sps > 10 ? 13 : 15);
    size_t nout = nbytes * 8 * (size_t)sps;
    if (!out) return nout;
    uint8_t* coded; size_t nc = tx_bits(bytes, nbytes, &coded);
    static const int map[4] = {0, 1, 3, 2};
    static const cf32 table[4] = {{-0.707f, -0.707f}, {-0.707f, 0.707f}, {0.707f, 0.707f}, {0.707f, -0.707f}};
    size_t ns = nc / 2;
    cf32* sym = NEW(cf32, ns);
    int prev = 0;
    for (size_t i = 0; i < ns; i++) {
        int v = (coded[2 * i] << 1) | coded[2 * i + 1];
        v = map[v];
        prev = (v + prev) % 4;
        sym[i] = table[prev];
    }
    free(coded);
    int nr = orc_root_raised_cosine(sps, sps, 1, 0.35, nfilts * sps, NULL);
    float* rrc = NEW(float, nr);
    orc_root_raised_cosine(sps, sps, 1, 0.35, nfilts * sps, rrc);
    size_t m = orc_resamp_ccf(sym, ns, rrc, nr, sps, 1, out);
    for (size_t i = 0; i < m; i++) { out[i].re *= 0.6f; out[i].im *= 0.6f; }
    free(rrc); free(sym);
    return m;
}

/* gr_mod_base.cpp:249-250: rational_resampler_ccf(fs/1e6, 1, low_pass(I, fs, 480k, 20k, BH)) */
size_t orc_tx_interp(const cf32* in, size_t n, int samp_rate, cf32* out)
{
    int I = samp_rate / 1000000;
    if (I < 2) { if (out) memcpy(out, in, n * sizeof(cf32)); return n; }
    if (!out) return n * (size_t)I;
    int nt = orc_low_pass(I, samp_rate, 480000, 20000, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* lp = NEW(float, nt);
    orc_low_pass(I, samp_rate, 480000, 20000, ORC_WIN_BLACKMAN_HARRIS, lp);
    size_t m = orc_resamp_ccf(in, n, lp, nt, I, 1, out);
    free(lp);
    return m;
}

static int16_t f2s(float x, float scale);
/* gr_demod_4fsk.cpp:28-200, FM branch (instances gr_demod_base.cpp:212-214,225: 4FSK2KFM sps 5, 4FSK1KFM sps 10, 4FSK10KFM
 * sps 1, 4FSK100K sps 2):  rational_resampler_ccf(I, D, low_pass(I, I*fs, T/2, T/2, BH)) -> fft_filter_ccf(low_pass(1, T, fw,
 * fw/2, BH)) [port 0] -> quadrature_demod_cf(sps/pi) -> fft_filter_fff(RRC(1.5, T, T/sps, 0.2, nfilts)) -> symbol_sync_ff(
 * TED_MOD_MUELLER_AND_MULLER, sps, 2pi/200, 1.0, 0.2869, 0.05, 1, constellation_rect{-1.5..1.5}) -> phase_modulator_fc(pi/2)
 * [port 1] -> complex_to_float -> interleave(4) with (imag, real) port order -> x128 +128 -> uchar -> cc_decoder -> descrambler
 * [port 2].  blocks::interleave(itemsize 4, blocksize 1)?  No: make(4) is the ITEM SIZE (bytes of a float), blocksize defaults
 * to 1, so the soft stream is im0, re0, im1, re1, ...  */
void orc_demod_4fsk(const cf32* in, size_t n, int sps, int samp_rate, int carrier_freq, int filter_width, int fm, orc_demod_out* o)
{
    (void)carrier_freq;
    memset(o, 0, sizeof *o);
    int target, sps_eff, decim, interp, nfilts;
    if (sps == 1)       { target = 80000;  sps_eff = 8;  decim = 25;  interp = 2; nfilts = 32 * 8; }
    else if (sps == 5)  { target = 20000;  sps_eff = 10; decim = 50;  interp = 1; nfilts = 25 * 10; }
    else if (sps == 10) { target = 10000;  sps_eff = 10; decim = 100; interp = 1; nfilts = 25 * 10; }
    else                { target = 500000; sps_eff = 5;  decim = 2;   interp = 1; nfilts = 50 * 5; }
    if ((nfilts % 2) == 0) nfilts += 1;
    int nt = orc_low_pass(interp, (double)interp * samp_rate, target / 2, target / 2, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* taps = NEW(float, nt);
    orc_low_pass(interp, (double)interp * samp_rate, target / 2, target / 2, ORC_WIN_BLACKMAN_HARRIS, taps);
    size_t n1 = orc_decim_count(n, interp, decim);
    cf32* s1 = NEW(cf32, n1);
    if (interp == 1) orc_decim_auto(in, n, taps, nt, decim, s1);
    else             orc_resamp_ccf(in, n, taps, nt, interp, decim, s1);
    free(taps);
    int nf = orc_low_pass(1, target, filter_width, filter_width / 2, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* ft = NEW(float, nf);
    orc_low_pass(1, target, filter_width, filter_width / 2, ORC_WIN_BLACKMAN_HARRIS, ft);
    o->filtered = NEW(cf32, n1); o->n_filtered = n1;
    orc_fir_ccf(s1, n1, ft, nf, o->filtered);
    free(ft); free(s1);
    if (!fm) {
        /* non-FM branch (ModemType4FSK2K, gr_demod_4fsk.cpp:52-60,110-127,165-181,186-189): four complex band-pass filters ->
         * complex_to_mag -> gr_4fsk_discriminator (strict arg-max, else 0) -> fft_filter_ccf(low_pass(1, T, T/sps, T/sps/20, BH))
         * -> symbol_sync_cc(MOD_M&M, sps, 2pi/200, 1.0, 0.2869, 0.05, 1, constellation_4fsk) [port 1] -> (re, im) interleaved */
        const int rs = sps == 1 ? 10000 : sps == 5 ? 2000 : 1000, bw = sps == 10 ? 2000 : 4000;
        const int fw = filter_width;
        const int lo_[4] = {-fw, -fw + rs, 0, fw - rs}, hi_[4] = {-fw + rs, 0, fw - rs, fw};
        int nb = orc_complex_band_pass(1, target, lo_[0], hi_[0], bw, ORC_WIN_BLACKMAN_HARRIS, NULL);
        float* mag[4];
        cf32* bt = NEW(cf32, nb); cf32* fo = NEW(cf32, n1);
        for (int q = 0; q < 4; q++) {
            orc_complex_band_pass(1, target, lo_[q], hi_[q], bw, ORC_WIN_BLACKMAN_HARRIS, bt);
            orc_fir_ccc(o->filtered, n1, bt, nb, fo);
            mag[q] = NEW(float, n1 + 1);
            for (size_t i = 0; i < n1; i++) mag[q][i] = sqrtf(fo[i].re * fo[i].re + fo[i].im * fo[i].im);
        }
        free(bt); free(fo);
        cf32* dsc = NEW(cf32, n1 + 1);
        const float A = (float)0.707107;
        for (size_t i = 0; i < n1; i++) {
            const float m1 = mag[0][i], m2 = mag[1][i], m3 = mag[2][i], m4 = mag[3][i];
            cf32 v = {0.0f, 0.0f};
            if (m1 > m2 && m1 > m3 && m1 > m4) { v.re = -A; v.im = -A; }
            else if (m2 > m1 && m2 > m3 && m2 > m4) { v.re = -A; v.im = A; }
            else if (m3 > m2 && m3 > m1 && m3 > m4) { v.re = A; v.im = A; }
            else if (m4 > m2 && m4 > m1 && m4 > m3) { v.re = A; v.im = -A; }
            dsc[i] = v;
        }
        for (int q = 0; q < 4; q++) free(mag[q]);
        int ns = orc_low_pass(1.0, target, target / sps_eff, target / sps_eff / 20, ORC_WIN_BLACKMAN_HARRIS, NULL);
        float* st = NEW(float, ns);
        orc_low_pass(1.0, target, target / sps_eff, target / sps_eff / 20, ORC_WIN_BLACKMAN_HARRIS, st);
        cf32* sf = NEW(cf32, n1 + 1);
        orc_fir_ccf(dsc, n1, st, ns, sf);
        free(st); free(dsc);
        cf32* sy = NEW(cf32, n1 / (size_t)(sps_eff - 1) + 16);
        size_t nsym = orc_symbol_sync_cc(sf, n1, ORC_TED_MOD_MM, (float)sps_eff, (float)(2 * M_PI / 200.0f), 1.0f, 0.2869f, 0.05f,
                                         ORC_CONST_4LEVEL, sy);
        free(sf);
        o->constellation = NEW(cf32, nsym + 1); o->n_const = nsym;
        float* fl2 = NEW(float, 2 * nsym + 2);
        for (size_t i = 0; i < nsym; i++) { o->constellation[i] = sy[i]; fl2[2 * i] = sy[i].re; fl2[2 * i + 1] = sy[i].im; }
        free(sy);
        uint8_t* soft2 = NEW(uint8_t, 2 * nsym + 2);
        orc_soft_quant(fl2, 2 * nsym, 128.0f, 128.0f, soft2);
        free(fl2);
        uint8_t* dec2 = NEW(uint8_t, nsym + 80);
        size_t nb2 = orc_cc_decode_k7(soft2, 2 * nsym, dec2);
        o->bits_a = NEW(uint8_t, nb2 + 1); o->n_bits_a = nb2;
        orc_descramble(dec2, nb2, 0x8A, 0x7F, 7, o->bits_a);
        free(soft2); free(dec2);
        return;
    }
    float* dem = NEW(float, n1);
    orc_quad_demod(o->filtered, n1, (float)(sps_eff / (1 * M_PI)), dem);
    int nr = orc_root_raised_cosine(1.5, target, target / sps_eff, 0.2, nfilts, NULL);
    float* rrc = NEW(float, nr);
    orc_root_raised_cosine(1.5, target, target / sps_eff, 0.2, nfilts, rrc);
    float* s5 = NEW(float, n1);
    orc_fir_fff(dem, n1, rrc, nr, s5);
    free(rrc); free(dem);
    float* sym = NEW(float, n1 / (size_t)(sps_eff - 1) + 16);
    size_t nsym = orc_symbol_sync_ff(s5, n1, ORC_TED_MOD_MM, (float)sps_eff, (float)(2 * M_PI / 200.0f), 1.0f, 0.2869f, 0.05f,
                                     ORC_CONST_4LEVEL, sym);
    free(s5);
    o->constellation = NEW(cf32, nsym); o->n_const = nsym;
    float* fl = NEW(float, 2 * nsym + 2);
    const float k = (float)(M_PI / 2);
    for (size_t i = 0; i < nsym; i++) {
        cf32 c; orc_sincosf(k * sym[i], &c.im, &c.re);
        o->constellation[i] = c;
        fl[2 * i] = c.im; fl[2 * i + 1] = c.re;       /* interleave: port 0 <- imag, port 1 <- real (gr_demod_4fsk.cpp:182-185) */
    }
    free(sym);
    uint8_t* soft = NEW(uint8_t, 2 * nsym + 2);
    orc_soft_quant(fl, 2 * nsym, 128.0f, 128.0f, soft);
    free(fl);
    uint8_t* dec = NEW(uint8_t, nsym + 80);
    size_t nb = orc_cc_decode_k7(soft, 2 * nsym, dec);
    o->bits_a = NEW(uint8_t, nb + 1); o->n_bits_a = nb;
    orc_descramble(dec, nb, 0x8A, 0x7F, 7, o->bits_a);
    free(soft); free(dec);
}

/* gr_mod_4fsk.cpp:28-115 (instances gr_mod_base.cpp:163-166,177) */
size_t orc_mod_4fsk(const uint8_t* bytes, size_t nbytes, int sps, int samp_rate, int carrier_freq, int filter_width, int fm, cf32* out)
{
    (void)carrier_freq;
    int nfilts = sps * 10, second_interp = 20;
    if (sps == 2) { sps = 5; second_interp = 2; nfilts = 256; }
    const int spacing = fm ? 1 : 2;
    const float amplif = fm ? 0.9f : 0.8f;
    size_t nout = nbytes * 8 * (size_t)sps * (size_t)second_interp;
    if (!out) return nout;
    uint8_t* coded; size_t nc = tx_bits(bytes, nbytes, &coded);
    static const int map[4] = {0, 1, 3, 2};
    static const float levels[4] = {-1.5f, -0.5f, 0.5f, 1.5f};
    size_t ns = nc / 2;
    float* sym = NEW(float, ns);
    for (size_t i = 0; i < ns; i++) sym[i] = levels[map[(coded[2 * i] << 1) | coded[2 * i + 1]]];
    free(coded);
    size_t n1 = ns * (size_t)sps;
    float* shaped = NEW(float, n1);
    if (fm) {
        int nr = orc_root_raised_cosine(sps, sps, 1, 0.2, nfilts, NULL);
        float* rrc = NEW(float, nr);
        orc_root_raised_cosine(sps, sps, 1, 0.2, nfilts, rrc);
        orc_resamp_fff(sym, ns, rrc, nr, sps, 1, shaped);
        free(rrc);
        const float sc = (float)0.66666666;
        for (size_t i = 0; i < n1; i++) shaped[i] = shaped[i] * sc;
    } else {
        for (size_t i = 0; i < n1; i++) shaped[i] = sym[i / (size_t)sps];
    }
    free(sym);
    cf32* fmv = NEW(cf32, n1);
    fm_mod(shaped, n1, (float)((spacing * M_PI) / sps), fmv);
    free(shaped);
    for (size_t i = 0; i < n1; i++) { fmv[i].re *= amplif; fmv[i].im *= amplif; }
    int nt = orc_low_pass(second_interp, samp_rate, filter_width, filter_width, ORC_WIN_HAMMING, NULL);
    float* lp = NEW(float, nt);
    orc_low_pass(second_interp, samp_rate, filter_width, filter_width, ORC_WIN_HAMMING, lp);
    size_t m = orc_resamp_ccf(fmv, n1, lp, nt, second_interp, 1, out);
    free(lp); free(fmv);
    return m;
}

/* gr_mod_m17 (reference src/gr/gr_mod_m17.cpp:26-81, instance make_gr_mod_m17() gr_mod_base.cpp:206 with the defaults of gr_mod_m17.h:43-44:
 * sps 125, 1 Msps, filter width 9000): bytes -> dibits (MSB first) -> map{2, 3, 1, 0} -> {-1.5, -0.5, 0.5, 1.5} ->
 * rational_resampler_fff(5, 1, RRC(5, 5, 1, 0.5, 250)) -> x0.66666666 -> frequency_modulator_fc(pi / 5) ->
 * fft_filter_ccf(low_pass(1, 24000, fw, fw, BH)) -> x0.9 -> x bb_gain -> rational_resampler_ccf(125, 3, low_pass(125, 3e6, 12000, 12000, BH)).
 * 2500 samples per 3 bytes. */
size_t orc_mod_m17(const uint8_t* bytes, size_t nbytes, int sps, int samp_rate, int filter_width, float bb_gain, cf32* out)
{
    const size_t ns = nbytes * 4, n24 = ns * 5, nout = orc_decim_count(n24, sps, 3);
    if (!out) return nout;
    static const int map[4] = {2, 3, 1, 0};
    static const float levels[4] = {-1.5f, -0.5f, 0.5f, 1.5f};
    float* sym = NEW(float, ns);
    for (size_t i = 0; i < ns; i++) sym[i] = levels[map[(bytes[i >> 2] >> (6 - 2 * (i & 3))) & 3]];
    int nr = orc_root_raised_cosine(5, 5, 1, 0.5, 250, NULL);
    float* rrc = NEW(float, nr);
    orc_root_raised_cosine(5, 5, 1, 0.5, 250, rrc);
    float* shaped = NEW(float, n24);
    orc_resamp_fff(sym, ns, rrc, nr, 5, 1, shaped);                              /* _first_resampler */
    free(rrc); free(sym);
    const float sc = (float)0.66666666;
    for (size_t i = 0; i < n24; i++) shaped[i] = shaped[i] * sc;                 /* _scale_pulses */
    cf32* fmv = NEW(cf32, n24);
    fm_mod(shaped, n24, (float)(M_PI / 5), fmv);                                 /* _fm_modulator */
    free(shaped);
    int nf = orc_low_pass(1, 24000, filter_width, filter_width, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* ft = NEW(float, nf);
    orc_low_pass(1, 24000, filter_width, filter_width, ORC_WIN_BLACKMAN_HARRIS, ft);
    cf32* g = NEW(cf32, n24);
    orc_fir_ccf(fmv, n24, ft, nf, g);                                            /* _filter */
    free(ft); free(fmv);
    for (size_t i = 0; i < n24; i++) { g[i].re *= 0.9f; g[i].im *= 0.9f; g[i].re *= bb_gain; g[i].im *= bb_gain; }   /* _amplify, _bb_gain */
    int nt = orc_low_pass(sps, (double)samp_rate * 3, 12000, 12000, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* lp = NEW(float, nt);
    orc_low_pass(sps, (double)samp_rate * 3, 12000, 12000, ORC_WIN_BLACKMAN_HARRIS, lp);
    const size_t m = orc_resamp_ccf(g, n24, lp, nt, sps, 3, out);                /* _resampler (125, 3) */
    free(lp); free(g);
    return m;
}

/* gr_zero_idle_bursts with delay > 0 (src/gr/gr_zero_idle_bursts.cpp:27-84), as ONE work() call over the whole stream sees it:
 *   - set_history(2 * SAMPLES_PER_SLOT) (:34-37; SAMPLES_PER_SLOT = 720, src/bursttimer.h:30) and work() copies in[i], the OLDEST item of
 *     the window: the stream leaves delayed by 2 * 720 - 1 = 1439 items (zeros first);
 *   - a tag {offset T, count} (re)loads the down-counter at OUTPUT item T - delay (`tag.offset == nitems + i + _delay`, :62-69): items
 *     T - delay .. T - delay + count - 1 leave as 0 + 0j; a later tag overwrites the counter.
 * (Across several work() calls the reference only looks at the tags inside the call's own window, :57, so a tag within the first `delay`
 *  items of a window is never matched: that part of its behaviour depends on how the scheduler cuts the stream and is not restated.)
 * runs = {ignored, T, count} triples, as orc_zero_idle_bursts takes them. */
void orc_zero_idle_bursts_delay(const cf32* in, size_t n, unsigned delay, const uint64_t* runs, size_t nruns, cf32* out)
{
    const size_t H1 = delay > 0 ? 2 * 720 - 1 : 0;
    uint64_t counter = 0;
    for (size_t i = 0; i < n; i++) {
        for (size_t r = 0; r < nruns; r++)
            if (runs[3 * r + 1] == (uint64_t)i + delay) { counter = runs[3 * r + 2]; break; }
        if (counter > 0) { out[i].re = 0.0f; out[i].im = 0.0f; counter--; }
        else if (i >= H1) out[i] = in[i - H1];
        else { out[i].re = 0.0f; out[i].im = 0.0f; }
    }
}

/* gr_mod_dmr (reference src/gr/gr_mod_dmr.cpp:26-90, instance make_gr_mod_dmr() gr_mod_base.cpp:207 with the defaults of gr_mod_dmr.h:38-39:
 * sps 125, 1 Msps, filter width 5000): bytes -> dibits (MSB first) -> map{2, 3, 1, 0} -> {-1.5, -0.5, 0.5, 1.5} ->
 * rational_resampler_fff(5, 1, RRC(5, 24000, 4800, 0.2, 125)) -> x0.66666666 -> frequency_modulator_fc(pi 4800 0.85 / 24000) ->
 * gr_zero_idle_bursts(delay = (125 - 1) / 2 = 62) -> x0.9 -> x bb_gain -> rational_resampler_ccf(125, 3, low_pass_2(125, 3e6, fw, 2000, 60, BH)).
 * (The fft_filter_ccf the constructor also creates, :74-75, is not connected.)  2500 samples per 3 bytes; zero_runs = {ignored, T, count}
 * triples in the zero-idle block's 24 ksps input coordinates (the "zero_samples" tags of gr_dmr_source once GNU Radio has scaled their
 * offsets through the chain), NULL / 0: none. */
size_t orc_mod_dmr(const uint8_t* bytes, size_t nbytes, int sps, int samp_rate, int filter_width, float bb_gain, const uint64_t* zero_runs, size_t nruns, cf32* out)
{
    const size_t ns = nbytes * 4, n24 = ns * 5, nout = orc_decim_count(n24, sps, 3);
    if (!out) return nout;
    static const int map[4] = {2, 3, 1, 0};
    static const float levels[4] = {-1.5f, -0.5f, 0.5f, 1.5f};
    float* sym = NEW(float, ns + 1);
    for (size_t i = 0; i < ns; i++) sym[i] = levels[map[(bytes[i >> 2] >> (6 - 2 * (i & 3))) & 3]];
    int nr = orc_root_raised_cosine(5, 24000, 4800, 0.2, 125, NULL);
    float* rrc = NEW(float, nr);
    orc_root_raised_cosine(5, 24000, 4800, 0.2, 125, rrc);
    float* shaped = NEW(float, n24 + 1);
    orc_resamp_fff(sym, ns, rrc, nr, 5, 1, shaped);                              /* _first_resampler */
    const unsigned delay = (unsigned)((nr - 1) / 2);                              /* first_resampler_delay, gr_mod_dmr.cpp:57 */
    free(rrc); free(sym);
    const float sc = (float)0.66666666;
    for (size_t i = 0; i < n24; i++) shaped[i] = shaped[i] * sc;                 /* _scale_pulses */
    cf32* fmv = NEW(cf32, n24 + 1);
    fm_mod(shaped, n24, (float)((M_PI * 4800.0 * 0.85) / 24000.0), fmv);         /* _fm_modulator (float sensitivity) */
    free(shaped);
    cf32* g = NEW(cf32, n24 + 1);
    orc_zero_idle_bursts_delay(fmv, n24, delay, zero_runs, nruns, g);            /* _zero_idle */
    free(fmv);
    for (size_t i = 0; i < n24; i++) { g[i].re *= 0.9f; g[i].im *= 0.9f; g[i].re *= bb_gain; g[i].im *= bb_gain; }   /* _amplify, _bb_gain */
    int nt = orc_low_pass_2(sps, (double)samp_rate * 3, filter_width, 2000, 60, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* lp = NEW(float, nt);
    orc_low_pass_2(sps, (double)samp_rate * 3, filter_width, 2000, 60, ORC_WIN_BLACKMAN_HARRIS, lp);
    const size_t m = orc_resamp_ccf(g, n24, lp, nt, sps, 3, out);                /* _resampler (125, 3) */
    free(lp); free(g);
    return m;
}

/* gr_mod_dsss (reference src/gr/gr_mod_dsss.cpp:27-92, instance make_gr_mod_dsss(25, 1000000, 1700, 150) gr_mod_base.cpp:170;
 * dsss_encoder_bb src/gr/dsss_encoder_bb_impl.cc:66-98): bytes -> bits -> scrambler -> K = 7 encoder -> every coded bit spread by the
 * Barker-13 code (bit 0: the code, bit 1: its complement) -> {-1, +1} -> rational_resampler_ccf(25, 1, RRC(25, 25, 1, 0.35, 275)) ->
 * x0.65 -> x bb_gain -> rational_resampler_ccf(50, 13, low_pass(50, 260000, fw, 5 fw)) -> rational_resampler_ccf(50, 1,
 * low_pass(50, 1e6, fw, 5 fw)).  1 000 000 samples per byte.  (The stream is complex with a zero imaginary part up to the first
 * resampler; it is carried as real here.) */
size_t orc_mod_dsss(const uint8_t* bytes, size_t nbytes, int sps, int samp_rate, int filter_width, float bb_gain, cf32* out)
{
    static const int barker_13[13] = {1, 1, 1, 1, 1, 0, 0, 1, 1, 0, 1, 0, 1};
    const size_t nchip = nbytes * 16 * 13, n52 = nchip * (size_t)sps, n20 = orc_decim_count(n52, 50, 13), nout = n20 * 50;
    if (!out) return nout;
    uint8_t* coded; size_t nc = tx_bits(bytes, nbytes, &coded);
    float* chips = NEW(float, nchip);
    for (size_t i = 0; i < nc; i++)
        for (int k = 0; k < 13; k++) chips[13 * i + k] = ((coded[i] ? ~barker_13[k] : barker_13[k]) & 1) ? 1.0f : -1.0f;
    free(coded);
    int nr = orc_root_raised_cosine(sps, sps, 1, 0.35, 11 * sps, NULL);
    float* rrc = NEW(float, nr);
    orc_root_raised_cosine(sps, sps, 1, 0.35, 11 * sps, rrc);
    float* shaped = NEW(float, n52);
    orc_resamp_fff(chips, nchip, rrc, nr, sps, 1, shaped);                       /* _resampler (25, 1) */
    free(rrc); free(chips);
    cf32* a = NEW(cf32, n52);
    for (size_t i = 0; i < n52; i++) { a[i].re = (shaped[i] * 0.65f) * bb_gain; a[i].im = 0.0f; }   /* _amplify, _bb_gain */
    free(shaped);
    int ni = orc_low_pass(50.0, 5200.0 * 50, filter_width, filter_width * 5, ORC_WIN_HAMMING, NULL);
    float* ti = NEW(float, ni);
    orc_low_pass(50.0, 5200.0 * 50, filter_width, filter_width * 5, ORC_WIN_HAMMING, ti);
    cf32* b = NEW(cf32, n20);
    orc_resamp_ccf(a, n52, ti, ni, 50, 13, b);                                   /* _resampler_if (50, 13) */
    free(ti); free(a);
    int nt = orc_low_pass(50, samp_rate, filter_width, filter_width * 5, ORC_WIN_HAMMING, NULL);
    float* tr = NEW(float, nt);
    orc_low_pass(50, samp_rate, filter_width, filter_width * 5, ORC_WIN_HAMMING, tr);
    const size_t m = orc_resamp_ccf(b, n20, tr, nt, 50, 1, out);                 /* _resampler_rf (50, 1) */
    free(tr); free(b);
    return m;
}

/* gr_mod_bpsk.cpp:28-67 (instances gr_mod_base.cpp:168-169: sps 500 / 250) */
size_t orc_mod_bpsk(const uint8_t* bytes, size_t nbytes, int sps, int samp_rate, int carrier_freq, int filter_width, cf32* out)
{
    (void)carrier_freq; (void)samp_rate; (void)filter_width;
    size_t nout = nbytes * 16 * (size_t)sps;
    if (!out) return nout;
    uint8_t* coded; size_t nc = tx_bits(bytes, nbytes, &coded);
    cf32* sym = NEW(cf32, nc);
    for (size_t i = 0; i < nc; i++) { sym[i].re = coded[i] ? 1.0f : -1.0f; sym[i].im = 0.0f; }
    free(coded);
    int nr = orc_root_raised_cosine(sps, sps, 1, 0.35, 11 * sps, NULL);
    float* rrc = NEW(float, nr);
    orc_root_raised_cosine(sps, sps, 1, 0.35, 11 * sps, rrc);
    size_t m = orc_resamp_ccf(sym, nc, rrc, nr, sps, 1, out);
    for (size_t i = 0; i < m; i++) { out[i].re *= 0.6f; out[i].im *= 0.6f; }
    free(rrc); free(sym);
    return m;
}

/* gr_demod_bpsk.cpp:36-103 (instances gr_demod_base.cpp:216-217: sps 10 / 5, both at 20 ksps):
 * rational_resampler_ccf(1, 50, low_pass(1, fs, 10k, 10k, BH)) -> fll_band_edge_cc(sps, 0.35, 32, 8pi/100) -> fft_filter_ccf(
 * RRC(sps, sps, 1, 0.35, 15 sps)) [port 0] -> agc2_cc(0.1, 0.1, 1, 1) -> clock_recovery_mm_cc(sps, 2.5e-5, 0.5, 0.05, 0.001)
 * -> costas_loop_cc(2pi/200, 2) [port 1] -> complex_to_real -> x64 +128 -> uchar -> 2x {cc_decoder -> descrambler} [ports 2, 3] */
void orc_demod_bpsk(const cf32* in, size_t n, int sps, int samp_rate, int carrier_freq, int filter_width, orc_demod_out* o)
{
    (void)carrier_freq; (void)filter_width;
    memset(o, 0, sizeof *o);
    const int target = 20000;
    int nt = orc_low_pass(1, samp_rate, target / 2, target / 2, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* taps = NEW(float, nt);
    orc_low_pass(1, samp_rate, target / 2, target / 2, ORC_WIN_BLACKMAN_HARRIS, taps);
    size_t n1 = orc_decim_count(n, 1, 50);
    cf32* s1 = NEW(cf32, n1);
    orc_decim_auto(in, n, taps, nt, 50, s1);
    free(taps);
    cf32* s2 = NEW(cf32, n1);
    orc_fll_band_edge(s1, n1, (float)sps, 0.35f, 32, (float)(8 * M_PI / 100), s2);
    free(s1);
    int nr = orc_root_raised_cosine(sps, sps, 1, 0.35, 15 * sps, NULL);
    float* rrc = NEW(float, nr);
    orc_root_raised_cosine(sps, sps, 1, 0.35, 15 * sps, rrc);
    o->filtered = NEW(cf32, n1); o->n_filtered = n1;
    orc_fir_ccf(s2, n1, rrc, nr, o->filtered);
    free(rrc); free(s2);
    cf32* s3 = NEW(cf32, n1);
    orc_agc2(o->filtered, n1, 1e-1f, 1e-1f, 1.0f, 1.0f, 65536.0f, s3);
    cf32* s4 = NEW(cf32, n1 / (size_t)(sps > 1 ? sps - 1 : 1) + 16);
    const float gain_omega = 0.005f;
    size_t nsym = orc_clock_recovery_mm_cc(s3, n1, (float)sps, gain_omega * gain_omega, 0.5f, 0.05f, 0.001f, s4);
    free(s3);
    o->constellation = NEW(cf32, nsym + 1); o->n_const = nsym;
    orc_costas(s4, nsym, (float)(2 * M_PI / 200), 2, 0, o->constellation);
    free(s4);
    float* re = NEW(float, nsym + 1);
    for (size_t i = 0; i < nsym; i++) re[i] = o->constellation[i].re;
    fec_tail(re, nsym, 64.0f, 1, o);
    free(re);
}

/* ------------------------------------------------------------------------------------------
 * DSSS mode "BPSK 8" (reference src/gr/gr_demod_dsss.cpp:30-111, instance make_gr_demod_dsss(25, 1000000, 1700, 150)
 * gr_demod_base.cpp:218; src/gr/dsss_decoder_cc_impl.cc:41-107 (matched-filter taps), :128-167 (general_work)).
 * ------------------------------------------------------------------------------------------ */
/* matched filter of dsss_decoder_cc: the Barker-13 code, reversed, `sps` samples per chip, through the RRC(1, sps, 1, 0.35,
 * 11 sps) pulse: taps[i] = sum_k rrc[k] cs[i + nr - 1 - k], i < 13 sps + 11 sps (fir_filter_ccf::filter over the zero-extended chip
 * sequence; one float fmaf chain, k ascending).  Real valued (the reference stores them as complex with zero imaginary part). */
int orc_dsss_taps(int sps, float* taps)
{
    static const int barker_13[13] = {1, 1, 1, 1, 1, 0, 0, 1, 1, 0, 1, 0, 1};
    const int rrc_ntaps = sps * 11, csz = 13 * sps, extra = rrc_ntaps, nt = csz + extra;
    if (!taps) return nt;
    int nr = orc_root_raised_cosine(1, sps, 1.0, (double)0.350f, rrc_ntaps, NULL);
    float* rrc = NEW(float, nr);
    orc_root_raised_cosine(1, sps, 1.0, (double)0.350f, rrc_ntaps, rrc);
    float* cs = NEW(float, csz + 2 * extra + nr);
    memset(cs, 0, sizeof(float) * (size_t)(csz + 2 * extra + nr));
    for (int i = 0; i < 13; i++)
        for (int k = 0; k < sps; k++) cs[extra + i * sps + k] = barker_13[13 - (i + 1)] == 0 ? -1.0f : 1.0f;
    for (int i = 0; i < nt; i++) {
        float a = 0.0f;
        for (int k = 0; k < nr; k++) a = fmaf(rrc[k], cs[i + nr - 1 - k], a);
        taps[i] = a;
    }
    free(rrc); free(cs);
    return nt;
}
/* dsss_decoder_cc::general_work: output I = the matched-filter value of largest magnitude among the 13 sps evaluations
 * j = 0 .. 13 sps - 1 over the windows x[13 sps (I - 2) + 1 + j .. + nt) (set_history(13 sps): in[0] of a work call is the item 13 sps - 1 before the first new one, and the block reads from in + (i - 1) 13 sps + j -- pinned against the reference block itself, tests/test_ref_blocks.py; the block looks one code period
 * back), first maximum wins, scaled by 2 / (13 sps).  Filter value = sum_k taps[k] x[P + nt - 1 - k], one fmaf chain per
 * component, k ascending; |v| = sqrtf(re^2 + im^2).  Output I exists once all of its windows do: n >= 13 sps (I - 1) + nt.
 * Returns the number of outputs. */
size_t orc_dsss_decoder(const cf32* in, size_t n, int sps, cf32* out)
{
    const int L = 13 * sps, nt = orc_dsss_taps(sps, NULL);
    float* taps = NEW(float, nt);
    orc_dsss_taps(sps, taps);
    /* last index of window j = L - 1 of output I: L (I - 2) + 1 + L - 1 + nt - 1 = L (I - 1) + nt - 1  =>  n >= L (I - 1) + nt */
    const long long need0 = (long long)nt - L;
    size_t nout = 0;
    if ((long long)n >= need0) nout = (size_t)(((long long)n - need0) / L) + 1;
    for (size_t I = 0; I < nout; I++) {
        float max_abs = 0.0f; cf32 max_val = {0.0f, 0.0f};
        for (int j = 0; j < L; j++) {
            const long long P = (long long)L * ((long long)I - 2) + 1 + j;
            float ar = 0.0f, ai = 0.0f;
            for (int k = 0; k < nt; k++) {
                const long long idx = P + nt - 1 - k;
                float xr = 0.0f, xi = 0.0f;
                if (idx >= 0) { xr = in[idx].re; xi = in[idx].im; }
                ar = fmaf(taps[k], xr, ar);
                ai = fmaf(taps[k], xi, ai);
            }
            const float a2 = ar * ar, b2 = ai * ai;
            const float cur = sqrtf(a2 + b2);
            if (cur > max_abs) { max_abs = cur; max_val.re = ar; max_val.im = ai; }
        }
        const float sc = 2.0f / (float)L;
        out[I].re = max_val.re * sc; out[I].im = max_val.im * sc;
    }
    free(taps);
    return nout;
}
void orc_demod_dsss(const cf32* in, size_t n, int sps, int samp_rate, int filter_width, orc_demod_out* o)
{
    memset(o, 0, sizeof *o);
    int nt = orc_low_pass(1, samp_rate, 10000, 10000, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* taps = NEW(float, nt);
    orc_low_pass(1, samp_rate, 10000, 10000, ORC_WIN_BLACKMAN_HARRIS, taps);
    size_t n1 = orc_decim_count(n, 1, 50);
    cf32* s1 = NEW(cf32, n1 + 1);
    orc_decim_auto(in, n, taps, nt, 50, s1);                                   /* _resampler (1, 50) -> 20 ksps */
    free(taps);
    int ni = orc_low_pass(1, 20000, 2600, 2600, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* ti = NEW(float, ni);
    orc_low_pass(1, 20000, 2600, 2600, ORC_WIN_BLACKMAN_HARRIS, ti);
    size_t n2 = orc_decim_count(n1, 13, 50);
    cf32* s2 = NEW(cf32, n2 + 1);
    orc_resamp_ccf(s1, n1, ti, ni, 13, 50, s2);                                /* _resampler_if (13, 50) -> 5200 sps */
    free(ti); free(s1);
    cf32* s3 = NEW(cf32, n2 + 1);
    orc_costas(s2, n2, (float)(M_PI / 200), 2, 1, s3);                         /* _costas_freq */
    free(s2);
    int nf = orc_low_pass(1, 5200, filter_width, 1200, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* ft = NEW(float, nf);
    orc_low_pass(1, 5200, filter_width, 1200, ORC_WIN_BLACKMAN_HARRIS, ft);
    o->filtered = NEW(cf32, n2 + 1); o->n_filtered = n2;
    orc_fir_ccf(s3, n2, ft, nf, o->filtered);                                  /* _filter -> port 0 */
    free(ft); free(s3);
    cf32* s5 = NEW(cf32, n2 + 1);
    orc_agc2(o->filtered, n2, 1e-1f, 1e-1f, 1.0f, 10.0f, 65536.0f, s5);        /* _agc (0.1, 0.1, 1, 10) */
    cf32* s6 = NEW(cf32, n2 / (size_t)(13 * sps) + 4);
    const size_t nd = orc_dsss_decoder(s5, n2, sps, s6);                       /* _dsss_decoder */
    free(s5);
    cf32* s7 = NEW(cf32, nd + 16);
    const float gain_omega = 0.005f;
    const size_t nsym = orc_clock_recovery_mm_cc(s6, nd, 1.0f, gain_omega * gain_omega, 0.5f, 0.05f, 0.005f, s7);
    free(s6);
    o->constellation = NEW(cf32, nsym + 1); o->n_const = nsym;
    orc_costas(s7, nsym, (float)(2 * M_PI / 100), 2, 0, o->constellation);     /* _costas_loop -> port 1 */
    free(s7);
    float* re = NEW(float, nsym + 1);
    for (size_t i = 0; i < nsym; i++) re[i] = o->constellation[i].re;
    fec_tail(re, nsym, 64.0f, 1, o);
    free(re);
}

/* single-carrier MMDVM receiver gr_demod_mmdvm.cpp:29-61 (instance make_gr_demod_mmdvm(): header defaults sps 10,
 * MMDVM_SAMPLE_RATE, 1700, filter_width 5000, gr_demod_mmdvm.h:35-36): rational_resampler_ccf(12, 125, low_pass_2(12, 12 fs, fw, 2000, 60, BH)) -> rssi_tag_block -> fft_filter_ccf(
 * low_pass_2(1, 24k, fw, 2000, 60, BH)) -> quadrature_demod_cf(24000/(2 pi 10000)) -> x1.0 -> float_to_short(1, 32767).
 * out: int16 [cap]; rssi: one dB value per 300 resampler outputs (may be NULL). Returns the sample count. */
size_t orc_demod_mmdvm(const cf32* in, size_t n, int samp_rate, int filter_width, int16_t* out, size_t cap, float* rssi, float cal,
                       size_t* n_rssi)
{
    int nt = orc_low_pass_2(12, 12.0 * samp_rate, filter_width, 2000, 60, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* taps = NEW(float, nt);
    orc_low_pass_2(12, 12.0 * samp_rate, filter_width, 2000, 60, ORC_WIN_BLACKMAN_HARRIS, taps);
    size_t n1 = orc_decim_count(n, 12, 125);
    cf32* a = NEW(cf32, n1 + 1);
    orc_resamp_ccf(in, n, taps, nt, 12, 125, a);
    free(taps);
    size_t nr = orc_rssi_tag(a, n1, cal, rssi);
    if (n_rssi) *n_rssi = nr;
    int nf = orc_low_pass_2(1, 24000, filter_width, 2000, 60, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* ft = NEW(float, nf);
    orc_low_pass_2(1, 24000, filter_width, 2000, 60, ORC_WIN_BLACKMAN_HARRIS, ft);
    cf32* b = NEW(cf32, n1 + 1);
    orc_fir_ccf(a, n1, ft, nf, b);
    float* d = NEW(float, n1 + 1);
    orc_quad_demod(b, n1, (float)(24000.0f / (2 * M_PI * 10000.0f)), d);
    size_t m = n1 < cap ? n1 : cap;
    for (size_t i = 0; i < m; i++) out[i] = f2s(d[i] * 1.0f, 32767.0f);
    free(a); free(b); free(d); free(ft);
    return m;
}

/* ------------------------------- batch driver for bench.py cpu_baseline ----------------------- */
static double now_s(void)
{
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
double orc_batch_rx(int mode, const cf32* iq, int batch, size_t n, int samp_rate, double carrier_offset_hz,
                    int threads, uint64_t* bit_checksum)
{
    uint64_t total = 0;
    double t0 = now_s();
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic) reduction(+ : total)
#endif
    for (int b = 0; b < batch; b++) {
        const cf32* x = iq + (size_t)b * n;
        size_t n1 = (samp_rate >= 2000000) ? orc_decim_count(n, 1, samp_rate / 1000000) : n;
        cf32* fe = NEW(cf32, n1);
        orc_frontend(x, n, samp_rate, carrier_offset_hz, fe);
        orc_demod_out o;
        if (mode == ORC_MODE_2FSK_1K)       orc_demod_2fsk(fe, n1, 10, 1000000, 1700, 2000, 0, &o);
        else if (mode == ORC_MODE_GMSK_10K) orc_demod_gmsk(fe, n1, 1, 1000000, 1700, 20000, &o);
        else                                orc_demod_qpsk(fe, n1, 2, 1000000, 1700, 160000, &o);
        uint64_t h = 1469598103934665603ull;
        for (size_t i = 0; i < o.n_bits_a; i++) h = (h ^ o.bits_a[i]) * 1099511628211ull;
        for (size_t i = 0; i < o.n_bits_b; i++) h = (h ^ o.bits_b[i]) * 1099511628211ull;
        total += h;
        orc_demod_out_free(&o);
        free(fe);
    }
    if (bit_checksum) *bit_checksum = total;
    return now_s() - t0;
}

/* ------------------------------- multi-carrier MMDVM RX (C4) ---------------------------------
 * gr_demod_mmdvm_multi2 (reference src/gr/gr_demod_mmdvm_multi2.cpp:58-135):
 *   stream_to_streams(M) -> pfb_channelizer_ccf(M, low_pass_2(1, fs, 5000, 2000, 60, BH), 1.0) ->
 *   per channel: rational_resampler_ccf(24, 25, low_pass_2(1, 600k, 5000, 2000, 60, BH)) ->
 *   fft_filter_ccf(low_pass_2(1, 24k, 5000, 2000, 60, BH)) -> quadrature_demod_cf(24000 / (2 pi 12500)) ->
 *   multiply_const_ff(1.0) -> float_to_short(1, 32767).
 * The reference fixes M = 10, fs = 250 ksps (src/config_mmdvm.h:4); M and fs = 25 kHz * M are parameters here
 * so that BASELINE's 64-channel extrapolation uses the same code.  Channel c is centred at +c*fs/M
 * (c > M/2: negative frequencies), the reference's port map {0,1,2,3,9,8,7} is the caller's business.
 *
 * pfb_channelizer_ccf [gr-filter/lib/pfb_channelizer_ccf_impl.cc, polyphase_filterbank.cc], oversample 1:
 *   branch p: taps h[p + M k];  v_p[n] = sum_k h[p + M k] * x[M n - p - M k]   (one fmaf chain, k ascending)
 *   channel c: y_c[n] = sum_{p=0}^{M-1} v_p[n] * W[(p c) mod M],  W = orc_chan_twiddles (exp(+j 2 pi q / M) rounded to float, exactly
 *   conjugate symmetric), as FOUR real fmaf chains over the branches, p ascending -- sa = sum W.re v.re, sb = sum W.im v.im,
 *   sc = sum W.im v.re, sd = sum W.re v.im -- and y = (sa - sb, sc + sd).  (Upstream runs an unnormalised backward FFTW of size M;
 *   any FFT factorisation rounds differently, so the direct sum is the contract.) */
int orc_chan_proto_taps(int M, float* taps)
{
    return orc_low_pass_2(1, 25000.0 * M, 5000, 2000, 60, ORC_WIN_BLACKMAN_HARRIS, taps);
}
/* DFT twiddles of the channelizer contract: W[q] = exp(+j 2 pi q / M) rounded to float for q <= M/2, with sin(pi) = 0 exactly
 * (libm returns 1.2e-16 for the rounded argument), and W[M - q] = conj(W[q]) above: an exactly conjugate-symmetric table, as a
 * twiddle generator that exploits the symmetry produces it.  Consequence used by the GPU kernel: the four summation chains of
 * bin M - c equal those of bin c with sb and sc negated, bit for bit. */
void orc_chan_twiddles(int M, cf32* W)
{
    for (int q = 0; q <= M / 2; q++) { W[q].re = (float)cos(2 * M_PI * q / M); W[q].im = (2 * q == M) ? 0.0f : (float)sin(2 * M_PI * q / M); }
    for (int q = M / 2 + 1; q < M; q++) { W[q].re = W[M - q].re; W[q].im = -W[M - q].im; }
}
size_t orc_pfb_channelizer(const cf32* in, size_t n, const float* taps, int nt, int M, cf32* out /* [M][n/M] */)
{
    orc_trace_event("pfb_channelizer(%d,%s,1)", M, orc_trace_name(taps, sizeof(float) * (size_t)nt));   /* critically sampled */
    const size_t nout = n / (size_t)M;
    cf32* W = NEW(cf32, M);
    orc_chan_twiddles(M, W);
    cf32* v = NEW(cf32, M);
    for (size_t m = 0; m < nout; m++) {
        for (int p = 0; p < M; p++) {
            float ar = 0.f, ai = 0.f;
            for (int k = 0; p + M * k < nt; k++) {
                long long idx = (long long)m * M - p - (long long)M * k;
                if (idx < 0) break;
                ar = fmaf(taps[p + M * k], in[idx].re, ar);
                ai = fmaf(taps[p + M * k], in[idx].im, ai);
            }
            v[p].re = ar; v[p].im = ai;
        }
        for (int c = 0; c < M; c++) {
            /* DFT summation contract: four real fmaf chains over the branches, p ascending, combined at the end -- what four
             * f32 MFMA accumulators (W.re*v.re, W.im*v.im, W.im*v.re, W.re*v.im) produce on the GPU */
            float sa = 0.f, sb = 0.f, sc = 0.f, sd = 0.f;
            for (int p = 0; p < M; p++) {
                const cf32 w = W[(int)(((long long)p * c) % M)];
                sa = fmaf(w.re, v[p].re, sa);
                sb = fmaf(w.im, v[p].im, sb);
                sc = fmaf(w.im, v[p].re, sc);
                sd = fmaf(w.re, v[p].im, sd);
            }
            out[(size_t)c * nout + m].re = sa - sb; out[(size_t)c * nout + m].im = sc + sd;
        }
    }
    free(W); free(v);
    return nout;
}
/* float_to_short(1, scale): rint, saturate (gr-blocks float_array_to_int-style) */
static int16_t f2s(float x, float scale)
{
    float r = rintf(x * scale);
    if (r > 32767.0f) r = 32767.0f;
    if (r < -32768.0f) r = -32768.0f;
    return (int16_t)r;
}
/* whole C4 RX chain; out: [M][cap] int16, returns samples per channel (cap must be >= n/M*24/25 + 2) */
size_t orc_demod_mmdvm_multi(const cf32* in, size_t n, int M, int16_t* out, size_t cap)
{
    return orc_demod_mmdvm_multi_rssi(in, n, M, out, cap, NULL, 0, 0.0f);
}
/* same + the rssi_tag_block between filter and discriminator (gr_demod_mmdvm_multi2.cpp:96,126-127): rssi[c*rcap + k] */
size_t orc_demod_mmdvm_multi_rssi(const cf32* in, size_t n, int M, int16_t* out, size_t cap, float* rssi, size_t rcap, float cal)
{
    return orc_demod_mmdvm_multi_4fsk(in, n, M, out, cap, rssi, rcap, cal, NULL, 0, NULL);
}
/* same + the 4FSK symbol tail of gr_demod_dmr (gr_demod_dmr.cpp:62-105) on every channel's filtered 24 ksps signal:
 * dibits[c*dcap + k] (two bits per symbol), ndib[c] = bits produced */
/* the per-channel chain behind the channelizer (gr_demod_mmdvm_multi2.cpp:60-63,80-92,106-131): ch[c][n1] at 25 ksps ->
 * rational_resampler_ccf(24, 25) -> fft_filter_ccf -> rssi_tag_block -> quadrature_demod_cf -> x1.0 -> float_to_short (+ 4FSK tail) */
static size_t multi_tail(const cf32* ch, int M, size_t n1, int16_t* out, size_t cap, float* rssi, size_t rcap, float cal,
                         uint8_t* dibits, size_t dcap, size_t* ndib)
{
    int nr = orc_low_pass_2(1, 600000, 5000, 2000, 60, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* rt = NEW(float, nr);
    orc_low_pass_2(1, 600000, 5000, 2000, 60, ORC_WIN_BLACKMAN_HARRIS, rt);
    int nf = orc_low_pass_2(1, 24000, 5000, 2000, 60, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* ft = NEW(float, nf);
    orc_low_pass_2(1, 24000, 5000, 2000, 60, ORC_WIN_BLACKMAN_HARRIS, ft);
    const size_t n2 = orc_decim_count(n1, 24, 25);
    const float gain = (float)(24000.0f / (2 * M_PI * 12500.0f));
    cf32* a = NEW(cf32, n2 + 1); cf32* b = NEW(cf32, n2 + 1); float* d = NEW(float, n2 + 1);
    size_t m = n2 < cap ? n2 : cap;
    for (int c = 0; c < M; c++) {
        orc_resamp_ccf(ch + (size_t)c * n1, n1, rt, nr, 24, 25, a);
        orc_fir_ccf(a, n2, ft, nf, b);
        if (rssi) {
            float* tmp = NEW(float, n2 / 300 + 1);
            size_t nr = orc_rssi_tag(b, n2, cal, tmp);
            for (size_t k = 0; k < nr && k < rcap; k++) rssi[(size_t)c * rcap + k] = tmp[k];
            free(tmp);
        }
        if (dibits) {
            float* fd = NEW(float, n2 + 1);
            orc_quad_demod(b, n2, (float)(24000 / (M_PI / 2 * 4800.0f)), fd);
            int nrr = orc_root_raised_cosine(1, 24000, 4800, 0.2, 125, NULL);
            float* rrc = NEW(float, nrr);
            orc_root_raised_cosine(1, 24000, 4800, 0.2, 125, rrc);
            float* ff = NEW(float, n2 + 1);
            orc_fir_fff(fd, n2, rrc, nrr, ff);
            float* sym = NEW(float, n2 / 4 + 16);
            size_t nsym = orc_symbol_sync_ff(ff, n2, ORC_TED_MM, 5.0f, (float)(2 * M_PI / 100.0f), 1.0f, 0.2869f, 0.06f, ORC_CONST_4LEVEL, sym);
            uint8_t* bits = NEW(uint8_t, 2 * nsym + 2);
            orc_4fsk_symbols_to_bits(sym, nsym, NULL, bits);
            size_t nb = 2 * nsym < dcap ? 2 * nsym : dcap;
            memcpy(dibits + (size_t)c * dcap, bits, nb);
            ndib[c] = nb;
            free(fd); free(rrc); free(ff); free(sym); free(bits);
        }
        orc_quad_demod(b, n2, gain, d);
        for (size_t i = 0; i < m; i++) out[(size_t)c * cap + i] = f2s(d[i] * 1.0f, 32767.0f);
    }
    free(a); free(b); free(d); free(rt); free(ft);
    return m;
}
/* the per-channel chains alone, on nch channel streams of n1 items at 25 ksps (ch[c * n1 + i]): what the owner rank of a
 * channel-sharded multi-GPU job runs on the samples the all-to-all delivered (tests/test_sharding.py) */
size_t orc_mmdvm_channel_tails(const cf32* ch, int nch, size_t n1, int16_t* out, size_t cap, float* rssi, size_t rcap, float cal,
                               uint8_t* dibits, size_t dcap, size_t* ndib)
{
    return multi_tail(ch, nch, n1, out, cap, rssi, rcap, cal, dibits, dcap, ndib);
}
size_t orc_demod_mmdvm_multi_4fsk(const cf32* in, size_t n, int M, int16_t* out, size_t cap, float* rssi, size_t rcap, float cal,
                                  uint8_t* dibits, size_t dcap, size_t* ndib)
{
    int nt = orc_chan_proto_taps(M, NULL);
    float* taps = NEW(float, nt);
    orc_chan_proto_taps(M, taps);
    const size_t n1 = n / (size_t)M;
    cf32* ch = NEW(cf32, (size_t)M * n1 + 1);
    orc_pfb_channelizer(in, n, taps, nt, M, ch);
    free(taps);
    const size_t m = multi_tail(ch, M, n1, out, cap, rssi, rcap, cal, dibits, dcap, ndib);
    free(ch);
    return m;
}
/* BASELINE.json configs[3] taken literally (SURVEY.md 8(d) "C4 freq-xlating"): N x 25 kHz channels out of ONE wideband input at
 * fs = 25 kHz * N, each channel through its own frequency-translating decimating FIR -- rotator_cc(2 pi (-25000) ct / fs),
 * ct = c (c <= N/2) | c - N, the channel map of the PFB form -- + rational_resampler_ccf(1, N, low_pass_2(1, fs, 5000, 2000, 60,
 * BH)) (the reference's way of writing a freq-xlating FIR: gr_demod_mmdvm_multi.cpp:62-66,89-96,111-112; the prototype is the one
 * gr_demod_mmdvm_multi2.cpp:58-60 designs for its channelizer), then the per-channel chain of gr_demod_mmdvm_multi2 and the 4FSK tail. */
size_t orc_demod_mmdvm_xlating_bank_4fsk(const cf32* in, size_t n, int N, int16_t* out, size_t cap, float* rssi, size_t rcap, float cal,
                                         uint8_t* dibits, size_t dcap, size_t* ndib)
{
    const double fs = 25000.0 * N;
    int nt = orc_chan_proto_taps(N, NULL);
    float* taps = NEW(float, nt);
    orc_chan_proto_taps(N, taps);
    const size_t n1 = orc_decim_count(n, 1, N);
    cf32* ch = NEW(cf32, (size_t)N * n1 + 1);
    cf32* rot = NEW(cf32, n + 1);
    for (int c = 0; c < N; c++) {
        const int ct = c <= N / 2 ? c : c - N;
        const float carrier_offset = -25000.0f;
        const uint64_t inc = orc_phase_inc_to_turn(2 * M_PI * carrier_offset * ct / (float)fs);
        orc_rotator(in, n, inc, 0, rot);
        orc_decim_xlating(rot, n, taps, nt, N, ch + (size_t)c * n1);
    }
    free(rot); free(taps);
    const size_t m = multi_tail(ch, N, n1, out, cap, rssi, rcap, cal, dibits, dcap, ndib);
    free(ch);
    return m;
}

/* legacy "freq-xlating" multi-carrier receiver gr_demod_mmdvm_multi.cpp:58-123: per channel i
 *   rotator_cc(2 pi (-separation) ct / fs), ct = i (i <= 3) | 3 - i   [more than 7 channels: i <= N/2 ? i : i - N]
 *   -> rational_resampler_ccf(1, D, low_pass(1, fs, fw, 3500, BH)) -> fft_filter_ccf(low_pass(1, 24k, fw, 3500, BH))
 *   -> rssi_tag_block -> quadrature_demod_cf(24000 / (2 pi 12500)) -> x1.0 -> float_to_short(1, 32767),  fs = 24 kHz * D.
 * out[c*cap + k] int16, rssi[c*rcap + k] (may be NULL).  Returns samples per channel. */
size_t orc_demod_mmdvm_xlating(const cf32* in, size_t n, int N, int separation, int D, int fw, int16_t* out, size_t cap,
                               float* rssi, size_t rcap, float cal)
{
    const double fs = 24000.0 * D;
    int nt = orc_low_pass(1, fs, fw, 3500, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* taps = NEW(float, nt);
    orc_low_pass(1, fs, fw, 3500, ORC_WIN_BLACKMAN_HARRIS, taps);
    int nf = orc_low_pass(1, 24000, fw, 3500, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* ft = NEW(float, nf);
    orc_low_pass(1, 24000, fw, 3500, ORC_WIN_BLACKMAN_HARRIS, ft);
    const size_t n2 = orc_decim_count(n, 1, D);
    const float gain = (float)(24000.0f / (2 * M_PI * 12500.0f));
    cf32* rot = NEW(cf32, n + 1); cf32* a = NEW(cf32, n2 + 1); cf32* b = NEW(cf32, n2 + 1); float* d = NEW(float, n2 + 1);
    const size_t m = n2 < cap ? n2 : cap;
    for (int c = 0; c < N; c++) {
        const int ct = N <= 7 ? (c > 3 ? 3 - c : c) : (c <= N / 2 ? c : c - N);
        const float carrier_offset = (float)(-separation);
        const uint64_t inc = orc_phase_inc_to_turn(2 * M_PI * carrier_offset * ct / (float)fs);
        if (ct != 0) orc_trace_event("rotator(%.17g)", 2 * M_PI * carrier_offset * ct / (float)fs);        /* the centre carrier has no rotator */
        orc_rotator(in, n, inc, 0, rot);
        orc_decim_xlating(rot, n, taps, nt, D, a);
        orc_fir_ccf(a, n2, ft, nf, b);
        if (rssi) {
            float* tmp = NEW(float, n2 / 300 + 1);
            size_t nr = orc_rssi_tag(b, n2, cal, tmp);
            for (size_t k = 0; k < nr && k < rcap; k++) rssi[(size_t)c * rcap + k] = tmp[k];
            free(tmp);
        }
        orc_quad_demod(b, n2, gain, d);
        for (size_t i = 0; i < m; i++) out[(size_t)c * cap + i] = f2s(d[i] * 1.0f, 32767.0f);
    }
    free(rot); free(a); free(b); free(d); free(taps); free(ft);
    return m;
}

/* multi-carrier MMDVM transmitter gr_mod_mmdvm_multi2.cpp:30-128: per channel short_to_float(1, 32767) -> x1.0 ->
 * frequency_modulator_fc(2 pi 12500 / 24000) -> fft_filter_ccf(low_pass_2(1, 24k, fw, 2000, 60, BH)) -> x0.8 ->
 * rational_resampler_ccf(25, 24, low_pass_2(25, 600k, fw, 2000, 60, BH)) -> [gr_zero_idle_bursts: tag driven, pass-through here]
 * -> pfb_synthesizer_ccf(10, low_pass_2(10, 250k, fw, 2000, 60, BH), false) ports {0,1,2,3,9,8,7}, idle ports zero -> x(1/N).
 * pfb_synthesizer_ccf [gr-filter/lib/pfb_synthesizer_ccf_impl.cc, twox = false]: per block n the M port samples go through an
 * UNNORMALISED inverse DFT V_i[n] = sum_p x_p[n] e^{+j 2 pi i p / M} (four real fmaf chains, p ascending, like the
 * channelizer's DFT), then branch i is filtered with the polyphase taps h[i + M j] over ITS OWN history V_i[n - j] and the M
 * branch outputs leave in order: out[n M + i].  in: int16 [N][n]; out: 10 * n25 samples, n25 = count of the 25/24 resampler. */
/* gr_zero_idle_bursts (src/gr/gr_zero_idle_bursts.cpp:45-84, delay 0) as absolute runs: zero_runs = {channel, start, count}
 * triples at the rate of the block's input (25 ksps behind the resampler here, gr_mod_mmdvm_multi2.cpp:108; 24 ksps behind the FM
 * modulator in gr_mod_mmdvm.cpp:57-58): items start .. start + count - 1 of that channel are replaced by 0 + 0j. */
static void apply_zero_runs(cf32* x, size_t n, int chan, const uint64_t* runs, size_t nruns)
{
    /* gr_zero_idle_bursts.cpp:53-80: one down-counter; the tag at an item's offset (re)loads it before the item is judged */
    uint64_t counter = 0;
    for (size_t i = 0; i < n; i++) {
        for (size_t r = 0; r < nruns; r++)
            if ((int)runs[3 * r] == chan && runs[3 * r + 1] == (uint64_t)i) { counter = runs[3 * r + 2]; break; }
        if (counter > 0) { x[i].re = 0.0f; x[i].im = 0.0f; counter--; }
    }
}
/* the block alone (delay 0): in -> out with the tagged runs {channel (ignored), offset, count} zeroed */
void orc_zero_idle_bursts(const cf32* in, size_t n, const uint64_t* runs, size_t nruns, cf32* out)
{
    memcpy(out, in, n * sizeof(cf32));
    uint64_t* r0 = (uint64_t*)malloc(sizeof(uint64_t) * 3 * (nruns + 1));
    for (size_t r = 0; r < nruns; r++) { r0[3 * r] = 0; r0[3 * r + 1] = runs[3 * r + 1]; r0[3 * r + 2] = runs[3 * r + 2]; }
    apply_zero_runs(out, n, 0, r0, nruns);
    free(r0);
}
static const uint64_t* g_zero_runs = NULL; static size_t g_zero_nruns = 0;
void orc_set_zero_runs(const uint64_t* runs, size_t nruns) { g_zero_runs = runs; g_zero_nruns = nruns; }   /* for the next orc_mod_mmdvm* call */
size_t orc_mod_mmdvm_multi(const int16_t* in, size_t n, int N, int filter_width, cf32* out)
{
    const int M = 10;
    const size_t n25 = orc_decim_count(n, 25, 24);
    if (!out) return n25 * (size_t)M;
    int nf = orc_low_pass_2(1, 24000, filter_width, 2000, 60, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* ft = NEW(float, nf);
    orc_low_pass_2(1, 24000, filter_width, 2000, 60, ORC_WIN_BLACKMAN_HARRIS, ft);
    int nr = orc_low_pass_2(25, 600000, filter_width, 2000, 60, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* rt = NEW(float, nr);
    orc_low_pass_2(25, 600000, filter_width, 2000, 60, ORC_WIN_BLACKMAN_HARRIS, rt);
    int ns = orc_low_pass_2(10, 250000, filter_width, 2000, 60, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* st = NEW(float, ns);
    orc_low_pass_2(10, 250000, filter_width, 2000, 60, ORC_WIN_BLACKMAN_HARRIS, st);
    cf32* port = (cf32*)calloc((size_t)M * (n25 + 1), sizeof(cf32));   /* [M][n25], idle ports stay zero */
    float* f = NEW(float, n + 1); cf32* a = NEW(cf32, n + 1); cf32* b = NEW(cf32, n + 1);
    int m = 1;
    for (int c = 0; c < N; c++) {
        for (size_t i = 0; i < n; i++) f[i] = ((float)in[(size_t)c * n + i] / 32767.0f) * 1.0f;
        fm_mod(f, n, (float)(2 * M_PI * 12500.0f / 24000.0f), a);
        orc_fir_ccf(a, n, ft, nf, b);
        for (size_t i = 0; i < n; i++) { b[i].re *= 0.8f; b[i].im *= 0.8f; }
        const int p = c <= 3 ? c : 10 - m++;
        orc_resamp_ccf(b, n, rt, nr, 25, 24, port + (size_t)p * n25);
        apply_zero_runs(port + (size_t)p * n25, n25, c, g_zero_runs, g_zero_nruns);
    }
    free(f); free(a); free(b); free(ft); free(rt);
    cf32 W[10];
    for (int q = 0; q < M; q++) { W[q].re = (float)cos(2 * M_PI * q / M); W[q].im = (float)sin(2 * M_PI * q / M); }
    const int J = (ns + M - 1) / M;
    cf32* V = (cf32*)calloc((size_t)M * (n25 + 1), sizeof(cf32));       /* V[i][blk] */
    for (size_t blk = 0; blk < n25; blk++)
        for (int i = 0; i < M; i++) {
            float sa = 0.f, sb = 0.f, sc = 0.f, sd = 0.f;
            for (int p = 0; p < M; p++) {
                const cf32 w = W[(i * p) % M], x = port[(size_t)p * n25 + blk];
                sa = fmaf(w.re, x.re, sa); sb = fmaf(w.im, x.im, sb); sc = fmaf(w.im, x.re, sc); sd = fmaf(w.re, x.im, sd);
            }
            V[(size_t)i * n25 + blk].re = sa - sb; V[(size_t)i * n25 + blk].im = sc + sd;
        }
    orc_trace_event("pfb_synthesizer(%d,%s,0)", M, orc_trace_name(st, sizeof(float) * (size_t)ns));   /* restated inline here, twox = false */
    const float lvl = 1.0f / (float)N;
    for (size_t blk = 0; blk < n25; blk++)
        for (int i = 0; i < M; i++) {
            float ar = 0.f, ai = 0.f;
            for (int j = 0; j < J; j++) {
                if ((size_t)j > blk) break;
                const int k = i + M * j;
                const float h = k < ns ? st[k] : 0.0f;
                const cf32 v = V[(size_t)i * n25 + (blk - (size_t)j)];
                ar = fmaf(h, v.re, ar); ai = fmaf(h, v.im, ai);
            }
            out[blk * M + i].re = ar * lvl; out[blk * M + i].im = ai * lvl;
        }
    free(port); free(V); free(st);
    return n25 * (size_t)M;
}

/* single-carrier MMDVM transmitter gr_mod_mmdvm.cpp:27-64: short_to_float(1, 32767) -> x1.0 -> frequency_modulator_fc(2 pi 12500
 * / 24000) -> [zero idle bursts: pass-through] -> fft_filter_ccf(low_pass_2(1, 24k, fw, 2000, 60, BH)) -> x0.8 -> x bb_gain ->
 * rational_resampler_ccf(125, 12, low_pass_2(125, 3e6, fw, 2000, 60, BH)): 24 ksps -> 250 ksps */
size_t orc_mod_mmdvm(const int16_t* in, size_t n, int filter_width, float bb_gain, cf32* out)
{
    const size_t nout = orc_decim_count(n, 125, 12);
    if (!out) return nout;
    int nf = orc_low_pass_2(1, 24000, filter_width, 2000, 60, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* ft = NEW(float, nf);
    orc_low_pass_2(1, 24000, filter_width, 2000, 60, ORC_WIN_BLACKMAN_HARRIS, ft);
    int nr = orc_low_pass_2(125, 125 * 24000.0, filter_width, 2000, 60, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* rt = NEW(float, nr);
    orc_low_pass_2(125, 125 * 24000.0, filter_width, 2000, 60, ORC_WIN_BLACKMAN_HARRIS, rt);
    float* f = NEW(float, n + 1); cf32* a = NEW(cf32, n + 1); cf32* b = NEW(cf32, n + 1);
    for (size_t i = 0; i < n; i++) f[i] = ((float)in[i] / 32767.0f) * 1.0f;
    fm_mod(f, n, (float)(2 * M_PI * 12500.0f / 24000.0f), a);
    apply_zero_runs(a, n, 0, g_zero_runs, g_zero_nruns);
    orc_fir_ccf(a, n, ft, nf, b);
    for (size_t i = 0; i < n; i++) { b[i].re *= 0.8f; b[i].im *= 0.8f; b[i].re *= bb_gain; b[i].im *= bb_gain; }
    size_t m = orc_resamp_ccf(b, n, rt, nr, 125, 12, out);
    free(f); free(a); free(b); free(ft); free(rt);
    return m;
}

/* ------------------------------- DMR / 4FSK symbol demodulator (a37) ---------------------------------
 * gr_demod_dmr (reference src/gr/gr_demod_dmr.cpp:36-105, instance make_gr_demod_dmr(5, 1000000) gr_demod_base.cpp:253):
 *   rational_resampler_ccf(3, 125, low_pass_2(3, 3e6, 5000, 2000, 60, BH)) -> [port 0] -> quadrature_demod_cf(24000/(pi/2*4800))
 *   -> fft_filter_fff(RRC(1, 24000, 4800, 0.2, 125)) -> symbol_sync_ff(TED_MUELLER_AND_MULLER, 5, 2pi/100, 1.0, 0.2869, 0.06, 1,
 *   constellation_rect{-1.5,-0.5,0.5,1.5}) -> multiply_const(0.9) -> phase_modulator_fc(pi/2) -> [port 1] -> complex_to_float ->
 *   interleave -> binary_slicer_fb -> pack_k_bits(2) -> map{3,1,2,0} -> unpack_k_bits(2) -> [port 2].
 * phase_modulator_fc's sincosf is the deterministic polynomial here (orc_sincosf). */
static void symbols_to_bits_scaled(const float* sym, size_t nsym, float scale, cf32* constellation, uint8_t* bits)
{
    static const int map[4] = {3, 1, 2, 0};
    const float k = (float)(M_PI / 2);
    for (size_t i = 0; i < nsym; i++) {
        const float s = sym[i] * scale;
        cf32 c; orc_sincosf(k * s, &c.im, &c.re);
        if (constellation) constellation[i] = c;
        const int v = ((c.re >= 0.0f) << 1) | (c.im >= 0.0f);
        const int m = map[v];
        bits[2 * i] = (uint8_t)((m >> 1) & 1); bits[2 * i + 1] = (uint8_t)(m & 1);
    }
}
void orc_4fsk_symbols_to_bits(const float* sym, size_t nsym, cf32* constellation, uint8_t* bits)
{
    symbols_to_bits_scaled(sym, nsym, 0.9f, constellation, bits);   /* gr_demod_dmr.cpp:73: _level_control 0.9 */
}
/* gr_demod_m17 (reference src/gr/gr_demod_m17.cpp:32-103, instance make_gr_demod_m17() gr_demod_base.cpp:252): the 4FSK symbol
 * demodulator of the M17 mode.  rational_resampler_ccf(3, 125, low_pass(3, 3 fs, 12k, 12k, BH)) -> fft_filter_ccf(low_pass(1, 24k, fw,
 * fw, BH)) [port 0] -> quadrature_demod_cf(5 / pi) -> fft_filter_fff(RRC(1.5, 24k, 4.8k, 0.5, 250)) -> symbol_sync_ff(MOD_M&M, 5,
 * 2 pi / (4800 / 50), 1.0, 0.2869, 500 / 4800, 1, constellation_rect 4 level) -> phase_modulator_fc(pi / 2) [port 1] -> slicer ->
 * pack(2) -> map{3,1,2,0} -> unpack(2) [port 2] */
void orc_demod_m17(const cf32* in, size_t n, int samp_rate, int filter_width, orc_demod_out* o)
{
    memset(o, 0, sizeof *o);
    int nt = orc_low_pass(3, (double)samp_rate * 3, 12000, 12000, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* taps = NEW(float, nt);
    orc_low_pass(3, (double)samp_rate * 3, 12000, 12000, ORC_WIN_BLACKMAN_HARRIS, taps);
    size_t n1 = orc_decim_count(n, 3, 125);
    cf32* r = NEW(cf32, n1);
    orc_resamp_ccf(in, n, taps, nt, 3, 125, r);
    free(taps);
    int nf = orc_low_pass(1, 24000, filter_width, filter_width, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* ft = NEW(float, nf);
    orc_low_pass(1, 24000, filter_width, filter_width, ORC_WIN_BLACKMAN_HARRIS, ft);
    o->filtered = NEW(cf32, n1); o->n_filtered = n1;
    orc_fir_ccf(r, n1, ft, nf, o->filtered);
    free(ft); free(r);
    float* d = NEW(float, n1);
    orc_quad_demod(o->filtered, n1, (float)(5 / M_PI), d);
    int nr = orc_root_raised_cosine(1.5, 24000, 4800, 0.5, 250, NULL);
    float* rrc = NEW(float, nr);
    orc_root_raised_cosine(1.5, 24000, 4800, 0.5, 250, rrc);
    float* f = NEW(float, n1);
    orc_fir_fff(d, n1, rrc, nr, f);
    free(rrc); free(d);
    float* sym = NEW(float, n1 / 4 + 16);
    size_t nsym = orc_symbol_sync_ff(f, n1, ORC_TED_MOD_MM, 5.0f, (float)(2 * M_PI / (4800.0f / 50)), 1.0f, 0.2869f, 500.0f / 4800.0f, ORC_CONST_4LEVEL, sym);
    free(f);
    o->constellation = NEW(cf32, nsym); o->n_const = nsym;
    o->bits_a = NEW(uint8_t, 2 * nsym); o->n_bits_a = 2 * nsym;
    symbols_to_bits_scaled(sym, nsym, 1.0f, o->constellation, o->bits_a);
    free(sym);
}
/* port 3 of gr_demod_dmr (gr_demod_dmr.cpp:94): the RRC-filtered discriminator output at 24 ksps, what gr_dmr_dmo_sink is fed;
 * same arithmetic as the first half of orc_demod_dmr below.  out may be NULL to size. */
size_t orc_demod_dmr_port3(const cf32* in, size_t n, int samp_rate, float* out)
{
    size_t n1 = orc_decim_count(n, 3, 125);
    if (!out) return n1;
    int nt = orc_low_pass_2(3, (double)samp_rate * 3, 5000, 2000, 60, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* taps = NEW(float, nt);
    orc_low_pass_2(3, (double)samp_rate * 3, 5000, 2000, 60, ORC_WIN_BLACKMAN_HARRIS, taps);
    cf32* r = NEW(cf32, n1);
    orc_resamp_ccf(in, n, taps, nt, 3, 125, r);
    free(taps);
    float* d = NEW(float, n1);
    orc_quad_demod(r, n1, (float)(24000 / (M_PI / 2 * 4800.0f)), d);
    int nr = orc_root_raised_cosine(1, 24000, 4800, 0.2, 125, NULL);
    float* rrc = NEW(float, nr);
    orc_root_raised_cosine(1, 24000, 4800, 0.2, 125, rrc);
    orc_fir_fff(d, n1, rrc, nr, out);
    free(rrc); free(d); free(r);
    return n1;
}
void orc_demod_dmr(const cf32* in, size_t n, int sps, int samp_rate, orc_demod_out* o)
{
    (void)sps;
    memset(o, 0, sizeof *o);
    int nt = orc_low_pass_2(3, (double)samp_rate * 3, 5000, 2000, 60, ORC_WIN_BLACKMAN_HARRIS, NULL);
    float* taps = NEW(float, nt);
    orc_low_pass_2(3, (double)samp_rate * 3, 5000, 2000, 60, ORC_WIN_BLACKMAN_HARRIS, taps);
    size_t n1 = orc_decim_count(n, 3, 125);
    o->filtered = NEW(cf32, n1); o->n_filtered = n1;
    orc_resamp_ccf(in, n, taps, nt, 3, 125, o->filtered);
    free(taps);
    float* d = NEW(float, n1);
    orc_quad_demod(o->filtered, n1, (float)(24000 / (M_PI / 2 * 4800.0f)), d);
    int nr = orc_root_raised_cosine(1, 24000, 4800, 0.2, 125, NULL);
    float* rrc = NEW(float, nr);
    orc_root_raised_cosine(1, 24000, 4800, 0.2, 125, rrc);
    float* f = NEW(float, n1);
    orc_fir_fff(d, n1, rrc, nr, f);
    free(rrc); free(d);
    float* sym = NEW(float, n1 / 4 + 16);
    size_t nsym = orc_symbol_sync_ff(f, n1, ORC_TED_MM, 5.0f, (float)(2 * M_PI / 100.0f), 1.0f, 0.2869f, 0.06f, ORC_CONST_4LEVEL, sym);
    free(f);
    o->constellation = NEW(cf32, nsym); o->n_const = nsym;
    o->bits_a = NEW(uint8_t, 2 * nsym); o->n_bits_a = 2 * nsym;
    orc_4fsk_symbols_to_bits(sym, nsym, o->constellation, o->bits_a);
    free(sym);
}
