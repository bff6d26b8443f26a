// firdes.cpp — see firdes.hpp.  Host only (no HIP).
#include "firdes.hpp"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <mutex>

namespace qrl {

static constexpr double kPi = 3.14159265358979323846;

static double window_attenuation(Window w)
{
    switch (w) {
    case WIN_HAMMING: return 53;
    case WIN_HANN: return 44;
    case WIN_BLACKMAN: return 74;
    case WIN_RECTANGULAR: return 21;
    case WIN_BLACKMAN_HARRIS: return 92;
    }
    return 53;
}

std::vector<float> window(Window type, int ntaps)
{
    std::vector<float> w(ntaps, 1.0f);
    if (type == WIN_RECTANGULAR) return w;
    double c[4] = {0, 0, 0, 0};
    if (type == WIN_HAMMING) { c[0] = 0.54; c[1] = 0.46; }
    else if (type == WIN_HANN) { c[0] = 0.5; c[1] = 0.5; }
    else if (type == WIN_BLACKMAN) { c[0] = 0.42; c[1] = 0.5; c[2] = 0.08; }
    else { c[0] = 0.35875; c[1] = 0.48829; c[2] = 0.14128; c[3] = 0.01168; }
    const double M = ntaps - 1;
    for (int n = 0; n < ntaps; ++n) {
        double a = 2.0 * kPi * n / M, b = 4.0 * kPi * n / M, d = 6.0 * kPi * n / M;
        w[n] = static_cast<float>(c[0] - c[1] * std::cos(a) + c[2] * std::cos(b) - c[3] * std::cos(d));
    }
    return w;
}

static int make_odd(int n) { return (n & 1) ? n : n + 1; }

int compute_ntaps(double fs, double tw, Window w)
{
    return make_odd(static_cast<int>(window_attenuation(w) * fs / (22.0 * tw)));
}
int compute_ntaps_windes(double fs, double tw, double atten_db)
{
    return make_odd(static_cast<int>(atten_db * fs / (22.0 * tw)));
}

static std::vector<float> windowed_sinc(double gain, double fs, double fc, int ntaps, Window wt)
{
    std::vector<float> taps(ntaps);
    const std::vector<float> w = window(wt, ntaps);
    const int M = (ntaps - 1) / 2;
    const double wc = 2 * kPi * fc / fs;
    taps[M] = static_cast<float>(wc / kPi * w[M]);
    for (int n = 1; n <= M; ++n) {
        // upstream evaluates both sides separately: sin(-x)/(-x) == sin(x)/x exactly
        taps[M + n] = static_cast<float>(std::sin(n * wc) / (n * kPi) * w[M + n]);
        taps[M - n] = static_cast<float>(std::sin(-n * wc) / (-n * kPi) * w[M - n]);
    }
    double dc = taps[M];
    for (int n = 1; n <= M; ++n) dc += 2 * taps[M + n];
    const double g = gain / dc;
    for (float& t : taps) t = static_cast<float>(t * g);
    return taps;
}

std::vector<float> low_pass(double gain, double fs, double fc, double tw, Window w)
{
    return windowed_sinc(gain, fs, fc, compute_ntaps(fs, tw, w), w);
}
std::vector<float> low_pass_2(double gain, double fs, double fc, double tw, double atten_db, Window w)
{
    return windowed_sinc(gain, fs, fc, compute_ntaps_windes(fs, tw, atten_db), w);
}

std::vector<float> band_pass_2(double gain, double fs, double lo, double hi, double tw, double atten_db, Window wt)
{
    const int ntaps = compute_ntaps_windes(fs, tw, atten_db);
    std::vector<float> taps(ntaps);
    const std::vector<float> w = window(wt, ntaps);
    const int M = (ntaps - 1) / 2;
    const double w0 = 2 * kPi * lo / fs, w1 = 2 * kPi * hi / fs;
    for (int n = -M; n <= M; ++n)
        taps[n + M] = n == 0 ? static_cast<float>((w1 - w0) / kPi * w[n + M])
                             : static_cast<float>((std::sin(n * w1) - std::sin(n * w0)) / (n * kPi) * w[n + M]);
    double centre = taps[M];                                   // unity gain at the band centre
    for (int n = 1; n <= M; ++n) centre += 2 * taps[n + M] * std::cos(n * (w0 + w1) * 0.5);
    const double g = gain / centre;
    for (float& t : taps) t = static_cast<float>(t * g);
    return taps;
}
static std::vector<std::complex<float>> rotate_prototype(const std::vector<float>& lp, double fs, double lo, double hi)
{
    const int ntaps = (int)lp.size();
    std::vector<std::complex<float>> taps(ntaps);
    const float freq = static_cast<float>(kPi * (hi + lo) / fs);
    float phase = (ntaps & 1) ? -freq * static_cast<float>(ntaps >> 1)
                              : static_cast<float>(-freq / 2.0 * ((1 + 2 * ntaps) >> 1));
    for (int i = 0; i < ntaps; ++i) {
        taps[i] = {static_cast<float>(lp[i] * std::cos(static_cast<double>(phase))),
                   static_cast<float>(lp[i] * std::sin(static_cast<double>(phase)))};
        phase += freq;
    }
    return taps;
}
std::vector<std::complex<float>> complex_band_pass_2(double gain, double fs, double lo, double hi, double tw, double atten_db, Window w)
{
    return rotate_prototype(low_pass_2(gain, fs, (hi - lo) / 2, tw, atten_db, w), fs, lo, hi);
}
void deemph_taps(int sample_rate, double tau, double a[2], double b[2])
{
    const double fs = (double)sample_rate;
    const double w_c = 1.0 / tau;
    const double w_ca = 2.0 * fs * (double)tanf((float)(w_c / (2.0 * fs)));   // the reference calls tanf
    const double k = -w_ca / (2.0 * fs);
    const double p1 = (1.0 + k) / (1.0 - k);
    const double b0 = -k / (1.0 - k);
    b[0] = b0; b[1] = b0 * 1.0;
    a[0] = 1.0; a[1] = -p1;
}
void preemph_taps(int sample_rate, double tau, double a[2], double b[2])
{
    const double fs = (double)sample_rate, fh = 0.925 * fs / 2.0;
    const double w_cl = 1.0 / tau, w_ch = 2.0 * kPi * fh;
    const double w_cla = 2.0 * fs * (double)tanf((float)(w_cl / (2.0 * fs)));   // (the reference calls tanf)
    const double w_cha = 2.0 * fs * (double)tanf((float)(w_ch / (2.0 * fs)));
    const double kl = -w_cla / (2.0 * fs), kh = -w_cha / (2.0 * fs);
    const double z1 = (1.0 + kl) / (1.0 - kl), p1 = (1.0 + kh) / (1.0 - kh), b0 = (1.0 - kl) / (1.0 - kh);
    const double g = std::fabs(1.0 - p1) / (b0 * std::fabs(1.0 - z1));           // unity gain at DC
    b[0] = g * b0 * 1.0; b[1] = g * b0 * -z1;
    a[0] = 1.0; a[1] = -p1;
}
std::vector<float> squelch_envelope(int ramp)
{
    std::vector<float> e((size_t)ramp + 1, 1.0f);
    for (int k = 0; ramp && k <= ramp; ++k) e[k] = (float)(0.5 - std::cos(kPi * (double)k / (double)ramp) / 2.0);
    return e;
}

std::vector<std::complex<float>> complex_band_pass(double gain, double fs, double lo, double hi, double tw, Window w)
{
    const int ntaps = compute_ntaps(fs, tw, w);
    const std::vector<float> lp = windowed_sinc(gain, fs, (hi - lo) / 2, ntaps, w);
    std::vector<std::complex<float>> taps(ntaps);
    const float freq = static_cast<float>(kPi * (hi + lo) / fs);
    float phase = (ntaps & 1) ? -freq * static_cast<float>(ntaps >> 1)
                              : static_cast<float>(-freq / 2.0 * ((1 + 2 * ntaps) >> 1));
    for (int i = 0; i < ntaps; ++i) {
        taps[i] = {static_cast<float>(lp[i] * std::cos(static_cast<double>(phase))),
                   static_cast<float>(lp[i] * std::sin(static_cast<double>(phase)))};
        phase += freq;
    }
    return taps;
}

// firdes::gaussian(gain, spb, bt, ntaps) (gr_mod_gmsk.cpp:68-70)
std::vector<float> gaussian(double gain, double spb, double bt, int ntaps)
{
    std::vector<float> taps((size_t)ntaps);
    double scale = 0;
    const double dt = 1.0 / spb;
    const double s = 1.0 / (std::sqrt(std::log(2.0)) / (2 * M_PI * bt));
    double t0 = -0.5 * ntaps;
    for (int i = 0; i < ntaps; i++) {
        t0++;
        const double ts = s * dt * t0;
        taps[i] = (float)std::exp(-0.5 * ts * ts);
        scale += taps[i];
    }
    for (int i = 0; i < ntaps; i++) taps[i] = (float)(taps[i] / scale * gain);
    return taps;
}

std::vector<float> root_raised_cosine(double gain, double fs, double symrate, double alpha, int ntaps)
{
    ntaps |= 1;
    const double spb = fs / symrate;
    std::vector<float> taps(ntaps);
    double scale = 0;
    for (int i = 0; i < ntaps; ++i) {
        const double xi = i - ntaps / 2;
        const double x1 = kPi * xi / spb;
        double x2 = 4 * alpha * xi / spb;
        double x3 = x2 * x2 - 1;
        double num, den;
        if (std::fabs(x3) >= 0.000001) {
            if (i != ntaps / 2) num = std::cos((1 + alpha) * x1) + std::sin((1 - alpha) * x1) / (4 * alpha * xi / spb);
            else                num = std::cos((1 + alpha) * x1) + (1 - alpha) * kPi / (4 * alpha);
            den = x3 * kPi;
        } else {
            if (alpha == 1) { taps[i] = -1; scale += taps[i]; continue; }
            x3 = (1 - alpha) * x1;
            x2 = (1 + alpha) * x1;
            num = (std::sin(x2) * (1 + alpha) * kPi - std::cos(x3) * ((1 - alpha) * kPi * spb) / (4 * alpha * xi) +
                   std::sin(x3) * spb * spb / (4 * alpha * xi * xi));
            den = -32 * kPi * alpha * alpha * xi / spb;
        }
        taps[i] = static_cast<float>(4 * alpha * num / den);
        scale += taps[i];
    }
    for (float& t : taps) t = static_cast<float>(t * gain / scale);
    return taps;
}

static double sinc(double x) { return x == 0.0 ? 1.0 : std::sin(kPi * x) / (kPi * x); }

void fll_band_edge_taps(float sps, float rolloff, int n, std::vector<std::complex<float>>& lower,
                        std::vector<std::complex<float>>& upper)
{
    const int M = static_cast<int>(std::rint(n / sps));
    std::vector<float> bb(n);
    float power = 0;
    for (int i = 0; i < n; ++i) {
        const float k = static_cast<float>(-M + i * 2.0 / sps);
        bb[i] = static_cast<float>(sinc(rolloff * k - 0.5) + sinc(rolloff * k + 0.5));
        power += bb[i];
    }
    const int N = static_cast<int>((n - 1.0) / 2.0);
    lower.assign(n, {});
    upper.assign(n, {});
    for (int i = 0; i < n; ++i) {
        const float tap = bb[i] / power;
        const float k = static_cast<float>((-N + i) / (2.0 * sps));
        const double a = 2.0 * kPi * (1 + rolloff) * k;
        // upstream: d_taps[n-1-i] = t(i), applied through a reversing FIR => coefficient of y[n-j] is t(j)
        lower[i] = {static_cast<float>(tap * std::cos(-a)), static_cast<float>(tap * std::sin(-a))};
        upper[i] = {static_cast<float>(tap * std::cos(a)), static_cast<float>(tap * std::sin(a))};
    }
}

void control_loop_gains(float bw, float& alpha, float& beta)
{
    const float damping = std::sqrt(2.0f) / 2.0f;
    const float denom = static_cast<float>(1.0 + 2.0 * damping * bw + bw * bw);
    alpha = (4 * damping * bw) / denom;
    beta = (4 * bw * bw) / denom;
}

void clock_loop_gains(float loop_bw, float zeta, float ted_gain, float& alpha, float& beta)
{
    const double wn = loop_bw, z = zeta, zw = z * wn, k1 = 2.0 / ted_gain;
    double cosx = 1.0;
    if (z > 1.0) cosx = std::cosh(wn * std::sqrt(z * z - 1.0));
    else if (z < 1.0) cosx = std::cos(wn * std::sqrt(1.0 - z * z));
    alpha = static_cast<float>(k1 * std::exp(-zw) * std::sinh(zw));
    beta = static_cast<float>(k1 * (1.0 - std::exp(-zw) * (std::sinh(zw) + cosx)));
}

// ---- tables: values are upstream's printed constants, regenerated ----
static float printed(double v, const char* fmt)
{
    char b[64];
    std::snprintf(b, sizeof b, fmt, v);
    return static_cast<float>(std::strtod(b, nullptr));
}

std::vector<float> atan_table()
{
    std::vector<float> t(257);
    for (int i = 0; i < 256; ++i) t[i] = printed(std::atan(i / 255.0), "%.6e");
    t[256] = t[255];
    return t;
}
std::vector<float> tanh_table()
{
    std::vector<float> t(256);
    for (int i = 0; i < 256; ++i) t[i] = printed(std::tanh((i - 128) / 64.0), "%.8f");
    return t;
}

// MMSE interpolator: c = R^-1 p for bandwidth B = 1/4, R_kl = sinc(2B(k-l)), p_k = sinc(2B(k-3-mu))
std::vector<float> mmse_table()
{
    std::vector<float> t(129 * 8);
    for (int imu = 0; imu <= 128; ++imu) {
        const double mu = imu / 128.0;
        double a[8][9];
        for (int k = 0; k < 8; ++k) {
            for (int l = 0; l < 8; ++l) a[k][l] = sinc(0.5 * (k - l));
            a[k][8] = sinc(0.5 * ((k - 3) - mu));
        }
        for (int c = 0; c < 8; ++c) {  // Gauss-Jordan with partial pivoting
            int piv = c;
            for (int r = c + 1; r < 8; ++r) if (std::fabs(a[r][c]) > std::fabs(a[piv][c])) piv = r;
            if (piv != c) for (int k = 0; k < 9; ++k) std::swap(a[c][k], a[piv][k]);
            for (int r = 0; r < 8; ++r) {
                if (r == c) continue;
                const double f = a[r][c] / a[c][c];
                for (int k = c; k < 9; ++k) a[r][k] -= f * a[c][k];
            }
        }
        for (int j = 0; j < 8; ++j) {
            float v = printed(a[7 - j][8] / a[7 - j][7 - j], "%.5e");
            if (imu == 0) v = (j == 4) ? 1.0f : 0.0f;
            if (imu == 128) v = (j == 3) ? 1.0f : 0.0f;
            t[imu * 8 + j] = v;
        }
    }
    return t;
}

uint64_t phase_inc_to_turn(double rad)
{
    double t = rad / (2.0 * kPi);
    t -= std::floor(t);
    if (t >= 1.0) t = 0.0;
    return static_cast<uint64_t>(t * 18446744073709551616.0);
}

void sincos_turn_host(uint64_t angle, float& s, float& c)
{
    const uint32_t a = static_cast<uint32_t>(angle >> 32);
    const uint32_t q = (a + 0x20000000u) >> 30;
    const int32_t r = static_cast<int32_t>(a - (q << 30));
    const float x = static_cast<float>(r) * 1.4629180792671596e-9f;
    const float z = x * x;
    float ps = std::fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = std::fmaf(z, ps, -1.6666654611e-1f);
    ps = std::fmaf(x * z, ps, x);
    float pc = std::fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = std::fmaf(z, pc, 4.166664568298827e-2f);
    pc = std::fmaf(z * z, pc, std::fmaf(z, -0.5f, 1.0f));
    switch (q & 3) {
    case 0: s = ps; c = pc; break;
    case 1: s = pc; c = -ps; break;
    case 2: s = -ps; c = -pc; break;
    default: s = -pc; c = ps; break;
    }
}

std::vector<float> dsss_matched_filter(int sps)
{
    static const int barker_13[13] = {1, 1, 1, 1, 1, 0, 0, 1, 1, 0, 1, 0, 1};
    const int rrc_ntaps = sps * 11, csz = 13 * sps, extra = rrc_ntaps, nt = csz + extra;
    const std::vector<float> rrc = root_raised_cosine(1, sps, 1.0, (double)0.350f, rrc_ntaps);   // `float excess_bw = 0.350f` in the reference (dsss_decoder_cc_impl.cc:77)
    const int nr = (int)rrc.size();
    std::vector<float> cs((size_t)(csz + 2 * extra + nr), 0.0f), taps((size_t)nt);
    for (int i = 0; i < 13; ++i)
        for (int k = 0; k < sps; ++k) cs[(size_t)(extra + i * sps + k)] = barker_13[13 - (i + 1)] == 0 ? -1.0f : 1.0f;
    for (int i = 0; i < nt; ++i) {
        float a = 0.0f;
        for (int k = 0; k < nr; ++k) a = std::fmaf(rrc[(size_t)k], cs[(size_t)(i + nr - 1 - k)], a);
        taps[(size_t)i] = a;
    }
    return taps;
}

}  // namespace qrl
