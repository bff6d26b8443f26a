/* orc_firdes.c — filter design of the oracle (TEST INFRASTRUCTURE; see orc.h).
 * Restates gr-filter/lib/firdes.cc and gr-fft/lib/window.cc of GNU Radio 3.10 [GR-MEM]:
 * arithmetic in double, taps stored as float at the points upstream stores them. */
#include "orc.h"
#include <math.h>
#include <stdlib.h>

static double max_attenuation(int win)
{
    switch (win) {
    case ORC_WIN_HAMMING: return 53;
    case ORC_WIN_HANN: return 44;
    case ORC_WIN_BLACKMAN: return 74;
    case ORC_WIN_RECTANGULAR: return 21;
    case ORC_WIN_BLACKMAN_HARRIS: return 92;
    default: return 53;
    }
}

/* window::build -> coswindow(): w[n] = c0 - c1 cos(2 pi n/M) + c2 cos(4 pi n/M) - c3 cos(6 pi n/M) */
int orc_window(int type, int ntaps, float* w)
{
    double c0, c1, c2 = 0, c3 = 0;
    switch (type) {
    case ORC_WIN_HAMMING: c0 = 0.54; c1 = 0.46; break;
    case ORC_WIN_HANN: c0 = 0.5; c1 = 0.5; break;
    case ORC_WIN_BLACKMAN: c0 = 0.42; c1 = 0.5; c2 = 0.08; break;
    case ORC_WIN_BLACKMAN_HARRIS: c0 = 0.35875; c1 = 0.48829; c2 = 0.14128; c3 = 0.01168; break;
    case ORC_WIN_RECTANGULAR: for (int n = 0; n < ntaps; n++) w[n] = 1.0f; return ntaps;
    default: return -1;
    }
    double M = (double)(ntaps - 1);
    for (int n = 0; n < ntaps; n++) {
        double v = c0 - c1 * cos(2.0 * M_PI * n / M) + c2 * cos(4.0 * M_PI * n / M) - c3 * cos(6.0 * M_PI * n / M);
        w[n] = (float)v;
    }
    return ntaps;
}

int orc_compute_ntaps(double fs, double tw, int win)
{
    int ntaps = (int)(max_attenuation(win) * fs / (22.0 * tw));
    if ((ntaps & 1) == 0) ntaps++;
    return ntaps;
}

/* firdes::compute_ntaps_windes: fred harris' rule  N = A*fs/(22*tw), made odd [GR-MEM] */
int orc_compute_ntaps_windes(double fs, double tw, double atten_db)
{
    int ntaps = (int)(atten_db * fs / (22.0 * tw));
    if ((ntaps & 1) == 0) ntaps++;
    return ntaps;
}

static void lowpass_core(double gain, double fs, double fc, int ntaps, int win, float* taps)
{
    float* w = (float*)malloc(sizeof(float) * (size_t)ntaps);
    orc_window(win, ntaps, w);
    int M = (ntaps - 1) / 2;
    double fwT0 = 2 * M_PI * fc / fs;
    for (int n = -M; n <= M; n++) {
        if (n == 0) taps[n + M] = (float)(fwT0 / M_PI * w[n + M]);
        else        taps[n + M] = (float)(sin(n * fwT0) / (n * M_PI) * w[n + M]);
    }
    double fmax = taps[0 + M];
    for (int n = 1; n <= M; n++) fmax += 2 * taps[n + M];
    gain /= fmax;
    for (int i = 0; i < ntaps; i++) taps[i] = (float)(taps[i] * gain);
    free(w);
}

int orc_low_pass(double gain, double fs, double fc, double tw, int win, float* taps)
{
    int ntaps = orc_compute_ntaps(fs, tw, win);
    if (taps) lowpass_core(gain, fs, fc, ntaps, win, taps);
    if (taps) orc_trace_taps(taps, sizeof(float) * (size_t)ntaps, "low_pass(%.17g,%.17g,%.17g,%.17g,%d)", gain, fs, fc, tw, win);
    return ntaps;
}

int orc_low_pass_2(double gain, double fs, double fc, double tw, double atten_db, int win, float* taps)
{
    int ntaps = orc_compute_ntaps_windes(fs, tw, atten_db);
    if (taps) lowpass_core(gain, fs, fc, ntaps, win, taps);
    if (taps) orc_trace_taps(taps, sizeof(float) * (size_t)ntaps, "low_pass_2(%.17g,%.17g,%.17g,%.17g,%.17g,%d)", gain, fs, fc, tw, atten_db, win);
    return ntaps;
}

/* firdes::band_pass_2 [GR-MEM]: windowed difference of two sincs, unity gain at the band centre */
int orc_band_pass_2(double gain, double fs, double lo, double hi, double tw, double atten_db, int win, float* taps)
{
    const double gain_arg = gain;
    int ntaps = orc_compute_ntaps_windes(fs, tw, atten_db);
    if (!taps) return ntaps;
    float* w = (float*)malloc(sizeof(float) * (size_t)ntaps);
    orc_window(win, ntaps, w);
    int M = (ntaps - 1) / 2;
    double fwT0 = 2 * M_PI * lo / fs, fwT1 = 2 * M_PI * hi / fs;
    for (int n = -M; n <= M; n++) {
        if (n == 0) taps[n + M] = (float)((fwT1 - fwT0) / M_PI * w[n + M]);
        else        taps[n + M] = (float)((sin(n * fwT1) - sin(n * fwT0)) / (n * M_PI) * w[n + M]);
    }
    double fmax = taps[0 + M];
    for (int n = 1; n <= M; n++) fmax += 2 * taps[n + M] * cos(n * (fwT0 + fwT1) * 0.5);
    gain /= fmax;
    for (int i = 0; i < ntaps; i++) taps[i] = (float)(taps[i] * gain);
    free(w);
    orc_trace_taps(taps, sizeof(float) * (size_t)ntaps, "band_pass_2(%.17g,%.17g,%.17g,%.17g,%.17g,%.17g,%d)", gain_arg, fs, lo, hi, tw, atten_db, win);
    return ntaps;
}

int orc_complex_band_pass(double gain, double fs, double lo, double hi, double tw, int win, cf32* taps)
{
    int ntaps = orc_compute_ntaps(fs, tw, win);
    if (!taps) return ntaps;
    float* lp = (float*)malloc(sizeof(float) * (size_t)ntaps);
    lowpass_core(gain, fs, (hi - lo) / 2, ntaps, win, lp);
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
    orc_trace_taps(taps, sizeof(cf32) * (size_t)ntaps, "complex_band_pass(%.17g,%.17g,%.17g,%.17g,%.17g,%d)", gain, fs, lo, hi, tw, win);
    return ntaps;
}

int orc_root_raised_cosine(double gain, double fs, double symrate, double alpha, int ntaps, float* taps)
{
    const int ntaps_arg = ntaps;
    ntaps |= 1;
    if (!taps) return ntaps;
    double spb = fs / symrate;
    double scale = 0;
    for (int i = 0; i < ntaps; i++) {
        double x1, x2, x3, num, den;
        double xindx = i - ntaps / 2;
        x1 = M_PI * xindx / spb;
        x2 = 4 * alpha * xindx / spb;
        x3 = x2 * x2 - 1;
        if (fabs(x3) >= 0.000001) {
            if (i != ntaps / 2) num = cos((1 + alpha) * x1) + sin((1 - alpha) * x1) / (4 * alpha * xindx / spb);
            else                num = cos((1 + alpha) * x1) + (1 - alpha) * M_PI / (4 * alpha);
            den = x3 * M_PI;
        } else {
            if (alpha == 1) { taps[i] = -1; scale += taps[i]; continue; }
            x3 = (1 - alpha) * x1;
            x2 = (1 + alpha) * x1;
            num = (sin(x2) * (1 + alpha) * M_PI - cos(x3) * ((1 - alpha) * M_PI * spb) / (4 * alpha * xindx) +
                   sin(x3) * spb * spb / (4 * alpha * xindx * xindx));
            den = -32 * M_PI * alpha * alpha * xindx / spb;
        }
        taps[i] = (float)(4 * alpha * num / den);
        scale += taps[i];
    }
    for (int i = 0; i < ntaps; i++) taps[i] = (float)(taps[i] * gain / scale);
    orc_trace_taps(taps, sizeof(float) * (size_t)ntaps, "root_raised_cosine(%.17g,%.17g,%.17g,%.17g,%d)", gain, fs, symrate, alpha, ntaps_arg);
    return ntaps;
}

int orc_gaussian(double gain, double spb, double bt, int ntaps, float* taps)
{
    if (!taps) return ntaps;
    double scale = 0;
    double dt = 1.0 / spb;
    double s = 1.0 / (sqrt(log(2.0)) / (2 * M_PI * bt));
    double t0 = -0.5 * ntaps;
    for (int i = 0; i < ntaps; i++) {
        t0++;
        double ts = s * dt * t0;
        taps[i] = (float)exp(-0.5 * ts * ts);
        scale += taps[i];
    }
    for (int i = 0; i < ntaps; i++) taps[i] = (float)(taps[i] / scale * gain);
    orc_trace_taps(taps, sizeof(float) * (size_t)ntaps, "gaussian(%.17g,%.17g,%.17g,%d)", gain, spb, bt, ntaps);
    return ntaps;
}

/* fll_band_edge_cc_impl::design_filter [GR-MEM]; taps are stored reversed exactly like upstream */
static double sinc_pi(double x) { return x == 0.0 ? 1.0 : sin(M_PI * x) / (M_PI * x); }
void orc_fll_taps(float sps, float rolloff, int n, cf32* lower, cf32* upper)
{
    int M = (int)rint(n / sps);
    float power = 0;
    float* bb = (float*)malloc(sizeof(float) * (size_t)n);
    for (int i = 0; i < n; i++) {
        float k = (float)(-M + i * 2.0 / sps);
        float tap = (float)(sinc_pi(rolloff * k - 0.5) + sinc_pi(rolloff * k + 0.5));
        power += tap;
        bb[i] = tap;
    }
    int N = (int)((n - 1.0) / 2.0);
    for (int i = 0; i < n; i++) {
        float tap = bb[i] / power;
        float k = (float)((-N + i) / (2.0 * sps));
        double a = 2.0 * M_PI * (1 + rolloff) * k;
        lower[n - i - 1].re = (float)(tap * cos(-a)); lower[n - i - 1].im = (float)(tap * sin(-a));
        upper[n - i - 1].re = (float)(tap * cos(a));  upper[n - i - 1].im = (float)(tap * sin(a));
    }
    free(bb);
}

/* blocks::control_loop::update_gains [GR-MEM], damping = sqrt(2)/2 */
void orc_control_loop_gains(float bw, float* alpha, float* beta)
{
    float damping = sqrtf(2.0f) / 2.0f;
    float denom = (float)(1.0 + 2.0 * damping * bw + bw * bw);
    *alpha = (4 * damping * bw) / denom;
    *beta = (4 * bw * bw) / denom;
}

/* digital::clock_tracking_loop::update_gains [GR-MEM] (computed in double, rounded once) */
void orc_clock_loop_gains(float loop_bw, float zeta, float ted_gain, float* alpha, float* beta)
{
    double omega_n_T = loop_bw, z = zeta;
    double zeta_omega_n_T = z * omega_n_T;
    double k1 = 2.0 / ted_gain;
    double cosx;
    if (z > 1.0)       cosx = cosh(omega_n_T * sqrt(z * z - 1.0));
    else if (z == 1.0) cosx = 1.0;
    else               cosx = cos(omega_n_T * sqrt(1.0 - z * z));
    *alpha = (float)(k1 * exp(-zeta_omega_n_T) * sinh(zeta_omega_n_T));
    *beta = (float)(k1 * (1.0 - exp(-zeta_omega_n_T) * (sinh(zeta_omega_n_T) + cosx)));
}
