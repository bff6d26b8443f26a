/* orc_side.c — TEST INFRASTRUCTURE (CPU oracle): the side outputs of gr_demod_base.
 *   orc_rssi_block      reference src/gr/rssi_block.cpp:31-44  (complex_to_mag_squared -> moving_average_ff(2000, 1, 2000) ->
 *                       single_pole_iir_filter_ff(0.04) -> nlog10_ff -> multiply_const_ff(10) -> add_const_ff(level))
 *   orc_power_spectrum  reference src/gr/rx_fft.cpp:83-96,126-127 (window, forward FFT, volk_32fc_s32f_power_spectrum_32f,
 *                       halves swapped as get_fft_data does)
 * parity unpinned: GNU Radio / VOLK are not installable here; [GR-MEM] the block semantics are restated from their published
 * behaviour (see qradiolink_amd/csrc/kernels_side.hip for the moving-average work()-call contract).  The FFT is a float64
 * radix-2 transform (the reference uses FFTW in float): the spectrum is compared within a tolerance, not bit for bit. */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* the same double-precision log2 as the device (qradiolink_amd/csrc/devmath.hpp det_log2f) */
float orc_det_log2f(float x)
{
    if (!(x > 0.0f)) return x == 0.0f ? -127.0f : NAN;
    if (x > 3.4028234e38f) return 127.0f;
    int adj = 0;
    if (x < 1.17549435e-38f) { x *= 18446744073709551616.0f; adj = -64; }
    unsigned bits; memcpy(&bits, &x, 4);
    int e = (int)(bits >> 23) - 127 + adj;
    unsigned mb = (bits & 0x007fffffu) | 0x3f800000u;
    float m; memcpy(&m, &mb, 4);
    if (m > 1.41421354f) { m *= 0.5f; e += 1; }
    const double md = (double)m;
    const double t = (md - 1.0) / (md + 1.0);
    const double t2 = t * t;
    double s = 2.0 / 13.0;
    s = s * t2 + 2.0 / 11.0;
    s = s * t2 + 2.0 / 9.0;
    s = s * t2 + 2.0 / 7.0;
    s = s * t2 + 2.0 / 5.0;
    s = s * t2 + 2.0 / 3.0;
    s = s * t2 + 2.0;
    s = s * t;
    return (float)((double)e + s * 1.4426950408889634);
}

/* whole stream at once; a fresh moving-average sum at every absolute multiple of 2000 outputs (max_iter) */
void orc_rssi_block(const cf32* in, size_t n, float level, float* out)
{
    float* p = (float*)malloc((n ? n : 1) * sizeof(float));
    for (size_t i = 0; i < n; i++) { const float a = in[i].re * in[i].re, c = in[i].im * in[i].im; p[i] = a + c; }
    float sum = 0.0f;
    double prev = 0.0;
    const float n_log2_10 = 1.0f / log2f(10.0f);
    for (size_t i = 0; i < n; i++) {
        if (i % 2000 == 0) {
            sum = 0.0f;
            for (int k = 1999; k >= 1; k--) sum += i >= (size_t)k ? p[i - k] : 0.0f;
        }
        sum += p[i];
        const float ma = sum * 1.0f;
        sum -= i >= 1999 ? p[i - 1999] : 0.0f;
        const double y = 0.04 * (double)ma + (1.0 - 0.04) * prev;
        prev = y;
        float v = orc_det_log2f((float)y) * n_log2_10;
        v = v * 10.0f;
        v = v + level;
        out[i] = v;
    }
    free(p);
}

static void fft_r2(double* re, double* im, size_t n)
{
    for (size_t i = 1, j = 0; i < n; i++) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { double t = re[i]; re[i] = re[j]; re[j] = t; t = im[i]; im[i] = im[j]; im[j] = t; }
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const double ang = -2.0 * M_PI / (double)len;
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; k++) {
                const double wr = cos(ang * (double)k), wi = sin(ang * (double)k);
                const size_t a = i + k, b = i + k + len / 2;
                const double xr = re[b] * wr - im[b] * wi, xi = re[b] * wi + im[b] * wr;
                re[b] = re[a] - xr; im[b] = im[a] - xi;
                re[a] += xr; im[a] += xi;
            }
    }
}
/* one FFT frame: in[n] * window[n] -> forward transform (n a power of two) -> 10 log10(|X / n|^2), halves swapped */
void orc_power_spectrum(const cf32* in, const float* window, size_t n, float* out)
{
    double* re = (double*)malloc(n * sizeof(double));
    double* im = (double*)malloc(n * sizeof(double));
    for (size_t i = 0; i < n; i++) { re[i] = (double)(in[i].re * window[i]); im[i] = (double)(in[i].im * window[i]); }
    fft_r2(re, im, n);
    const float inv = 1.0f / (float)n;
    for (size_t i = 0; i < n; i++) {
        const float r = (float)re[i] * inv, q = (float)im[i] * inv;
        const float a = r * r, c = q * q;
        const float v = 3.01029995663981209120f * orc_det_log2f(a + c);
        out[i < n / 2 ? i + n / 2 : i - n / 2] = v;
    }
    free(re); free(im);
}
