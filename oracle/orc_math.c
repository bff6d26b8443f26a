/* orc_math.c — deterministic math primitives + tables of the oracle (TEST INFRASTRUCTURE).
 * See orc.h for the arithmetic contract.  [GR-MEM] = restated from GNU Radio 3.10 upstream. */
#include "orc.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

/* ---- polynomial sin/cos on [-pi/4, pi/4] (Cephes single-precision coefficients) ---- */
static inline float poly_sin(float x)
{
    float z = x * x;
    float p = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    p = fmaf(z, p, -1.6666654611e-1f);
    return fmaf(x * z, p, x);
}
static inline float poly_cos(float x)
{
    float z = x * x;
    float p = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    p = fmaf(z, p, 4.166664568298827e-2f);
    float q = fmaf(z, -0.5f, 1.0f);
    return fmaf(z * z, p, q);
}
static inline void quad_fix(int q, float ps, float pc, float* s, float* c)
{
    switch (q & 3) {
    case 0: *s = ps;  *c = pc;  break;
    case 1: *s = pc;  *c = -ps; break;
    case 2: *s = -ps; *c = -pc; break;
    default:*s = -pc; *c = ps;  break;
    }
}

/* sin/cos of a float angle (radians).  Cody-Waite reduction by pi/2 with a 2-term constant. */
void orc_sincosf(float x, float* s, float* c)
{
    const float TWO_OVER_PI = 0.636619772367581343f;
    const float PIO2_HI = 1.57079637050628662109375f;
    const float PIO2_LO = -4.37113900018624283e-8f;
    float k = rintf(x * TWO_OVER_PI);
    float r = fmaf(-k, PIO2_HI, x);
    r = fmaf(-k, PIO2_LO, r);
    quad_fix((int)k, poly_sin(r), poly_cos(r), s, c);
}

/* sin/cos of a 64-bit fixed-point angle, unit 2^-64 turn.  Uses the top 32 bits. */
void orc_sincos_turn(uint64_t angle, float* s, float* c)
{
    uint32_t a = (uint32_t)(angle >> 32);
    uint32_t q = (a + 0x20000000u) >> 30;                 /* nearest quadrant */
    int32_t  r = (int32_t)(a - (q << 30));                /* [-2^29, 2^29) */
    float x = (float)r * 1.4629180792671596e-9f;          /* 2*pi / 2^32 */
    quad_fix((int)q, poly_sin(x), poly_cos(x), s, c);
}

/* ---- tables ---- */
static float g_atan[257];
static float g_tanh[256];
static float g_mmse[129 * 8];
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

static float round_sig(double v, const char* fmt)
{
    char buf[64];
    snprintf(buf, sizeof buf, fmt, v);
    return (float)strtod(buf, NULL);
}

static double sinc_pi(double x) { return x == 0.0 ? 1.0 : sin(M_PI * x) / (M_PI * x); }

static void solve8(double A[8][9])
{
    for (int c = 0; c < 8; c++) {
        int piv = c;
        for (int r = c + 1; r < 8; r++) if (fabs(A[r][c]) > fabs(A[piv][c])) piv = r;
        if (piv != c) for (int k = 0; k < 9; k++) { double t = A[c][k]; A[c][k] = A[piv][k]; A[piv][k] = t; }
        for (int r = 0; r < 8; r++) {
            if (r == c) continue;
            double f = A[r][c] / A[c][c];
            for (int k = c; k < 9; k++) A[r][k] -= f * A[c][k];
        }
    }
    for (int r = 0; r < 8; r++) A[r][8] /= A[r][r];
}

static void init_tables(void)
{
    /* fast_atan_table [GR-MEM gnuradio-runtime/lib/math/fast_atan2f.cc]: atan(i/255) printed %e */
    for (int i = 0; i < 256; i++) g_atan[i] = round_sig(atan((double)i / 255.0), "%.6e");
    g_atan[256] = g_atan[255];
    /* tanh_lut_table [GR-MEM gr-blocks control_loop.h]: tanh((i-128)/64) printed with 8 decimals */
    for (int i = 0; i < 256; i++) g_tanh[i] = round_sig(tanh((double)(i - 128) / 64.0), "%.8f");
    /* MMSE interpolator taps [GR-MEM gr-filter/lib/interpolator_taps.h]: least-squares design for
     * one-sided bandwidth B = 0.25, 8 taps, 128 steps, printed %.5e. */
    const double B = 0.25;
    for (int imu = 0; imu <= 128; imu++) {
        double mu = imu / 128.0;
        double A[8][9];
        for (int k = 0; k < 8; k++) {
            for (int l = 0; l < 8; l++) A[k][l] = sinc_pi(2 * B * (double)(k - l));
            A[k][8] = sinc_pi(2 * B * ((double)(k - 3) - mu));
        }
        solve8(A);
        for (int j = 0; j < 8; j++) {
            double v = A[7 - j][8];
            float f = round_sig(v, "%.5e");
            if (imu == 0)   f = (j == 4) ? 1.0f : 0.0f;
            if (imu == 128) f = (j == 3) ? 1.0f : 0.0f;
            g_mmse[imu * 8 + j] = f;
        }
    }
}
const float* orc_atan_table(void) { pthread_once(&g_once, init_tables); return g_atan; }
const float* orc_tanh_table(void) { pthread_once(&g_once, init_tables); return g_tanh; }
const float* orc_mmse_table(void) { pthread_once(&g_once, init_tables); return g_mmse; }

/* [GR-MEM fast_atan2f.cc] table-lookup atan2 with linear interpolation, no fused ops */
float orc_fast_atan2f(float y, float x)
{
    const float* T = orc_atan_table();
    float y_abs = fabsf(y), x_abs = fabsf(x);
    if (!((y_abs > 0.0f) || (x_abs > 0.0f))) return 0.0f;
    float z = (y_abs < x_abs) ? (y_abs / x_abs) : (x_abs / y_abs);
    float base;
    if ((double)z < 0.003921569) {
        base = z;
    } else {
        float alpha = z * 255.0f;
        int index = ((int)alpha) & 0xff;
        alpha -= (float)index;
        float d = T[index + 1] - T[index];
        base = T[index] + d * alpha;
    }
    float angle;
    const float PI_F = 3.14159265358979323846f, PIO2_F = 1.57079632679489661923f;
    if (x_abs > y_abs) {
        if (x >= 0.0f) angle = (y >= 0.0f) ? base : -base;
        else           angle = (y >= 0.0f) ? (PI_F - base) : (base - PI_F);
    } else {
        if (y >= 0.0f) angle = (x >= 0.0f) ? (PIO2_F - base) : (PIO2_F + base);
        else           angle = (x >= 0.0f) ? (-PIO2_F + base) : (-PIO2_F - base);
    }
    return angle;
}

/* [GR-MEM control_loop.h tanhf_lut]; index clamped to 255 (upstream reads [256] at x==2) */
float orc_tanhf_lut(float x)
{
    const float* T = orc_tanh_table();
    if (x > 2.0f) return 1.0f;
    if (x <= -2.0f) return -1.0f;
    int index = (int)(128.0f + 64.0f * x);
    if (index > 255) index = 255;
    if (index < 0) index = 0;
    return T[index];
}
