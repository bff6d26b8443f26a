/* orc_blocks.c — stream blocks of the oracle (TEST INFRASTRUCTURE; see orc.h).
 * Each function restates one GNU Radio 3.10 block [GR-MEM] over a finite stream with the
 * block's zero initial state.  "exists iff" rules give the output counts of an infinite
 * stream truncated after n inputs, which is what makes results chunk-size independent. */
#include "orc.h"
#include "../include/qrl_contracts.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * rotator_cc  [gr-blocks/lib/rotator_cc_impl.cc, volk_32fc_s32fc_x2_rotator_32fc]
 * out[n] = in[n] * e^{j angle(n)}, angle(n) = acc0 + n*inc exact in 2^-64 turns (drift-free NCO;
 * VOLK's float recurrence with renormalisation every 512 samples drifts, ours does not).
 * phase(n) = T_hi(n>>9) (x) T_lo(n&511) with T_hi(b) = sincos_turn(acc0 + 512*b*inc),
 * T_lo(r) = sincos_turn(r*inc); both complex products use the fmaf pattern below.
 * ------------------------------------------------------------------------------------------ */
uint64_t orc_phase_inc_to_turn(double rad)
{
    double t = rad / (2.0 * M_PI);
    t -= floor(t);
    if (t >= 1.0) t = 0.0;
    return (uint64_t)(t * 18446744073709551616.0);
}
static inline cf32 cmul_fma(cf32 a, cf32 b)
{
    cf32 r;
    r.re = fmaf(a.re, b.re, -(a.im * b.im));
    r.im = fmaf(a.re, b.im, a.im * b.re);
    return r;
}
void orc_rotator(const cf32* in, size_t n, uint64_t inc, uint64_t acc0, cf32* out)
{
    cf32 lo[512];
    for (int r = 0; r < 512; r++) orc_sincos_turn((uint64_t)r * inc, &lo[r].im, &lo[r].re);
    cf32 hi = {1, 0};
    for (size_t i = 0; i < n; i++) {
        if ((i & 511) == 0) orc_sincos_turn(acc0 + (uint64_t)i * inc, &hi.im, &hi.re);
        cf32 ph = cmul_fma(hi, lo[i & 511]);
        out[i] = cmul_fma(in[i], ph);
    }
}

/* ------------------------------------------------------------------------------------------
 * rational_resampler_ccf / fff  [gr-filter/lib/rational_resampler_impl.cc]
 * y[i] = sum_j h[(i*D mod I) + j*I] * x[floor(i*D/I) - j];  y[i] exists iff floor(i*D/I) <= n-1.
 * ------------------------------------------------------------------------------------------ */
size_t orc_decim_count(size_t n, int interp, int decim)
{
    if (n == 0) return 0;
    /* largest i with floor(i*D/I) <= n-1  <=>  i*D <= (n-1)*I + I-1 */
    return (size_t)((((uint64_t)(n - 1) * (uint64_t)interp + (uint64_t)interp - 1) / (uint64_t)decim) + 1);
}

/* Pure decimator (I = 1).  Summation order (the contract the HIP kernel follows):
 * k = p + j*D, p in [0,D), j in [0,J).  The D phases are split into nsplit contiguous groups
 * g: p in [floor(g*D/G), floor((g+1)*D/G)); each group is ONE fmaf chain, p ascending then
 * j ascending, starting from +0;  y = (r0 + r1) + (r2 + r3)   (G=2: r0+r1, G=1: r0). */
size_t orc_decim_fir_ccf(const cf32* in, size_t n, const float* taps, int nt, int D, int G, cf32* out)
{
    size_t nout = orc_decim_count(n, 1, D);
    int J = (nt + D - 1) / D;
    for (size_t m = 0; m < nout; m++) {
        float rr[4] = {0, 0, 0, 0}, ri[4] = {0, 0, 0, 0};
        for (int g = 0; g < G; g++) {
            int p0 = (int)(((long)g * D) / G), p1 = (int)(((long)(g + 1) * D) / G);
            float ar = 0.0f, ai = 0.0f;
            for (int p = p0; p < p1; p++) {
                for (int j = 0; j < J; j++) {
                    int k = p + j * D;
                    if (k >= nt) break;
                    long long idx = (long long)(m - (size_t)0) * D - (long long)j * D - p;
                    if (idx < 0) break;
                    float h = taps[k];
                    ar = fmaf(h, in[idx].re, ar);
                    ai = fmaf(h, in[idx].im, ai);
                }
            }
            rr[g] = ar; ri[g] = ai;
        }
        if (G == 4)      { out[m].re = (rr[0] + rr[1]) + (rr[2] + rr[3]); out[m].im = (ri[0] + ri[1]) + (ri[2] + ri[3]); }
        else if (G == 2) { out[m].re = rr[0] + rr[1]; out[m].im = ri[0] + ri[1]; }
        else             { out[m].re = rr[0]; out[m].im = ri[0]; }
    }
    return nout;
}

/* "m16" summation contract of the wide decimators (the order an f32 MFMA 16x16x4 accumulation
 * produces: gfx950's f32-input MFMA is bit-for-bit a k-ordered fmaf chain).
 * Output m = 16a + b (b = m mod 16, absolute index).  With u = sample offset relative to the
 * block origin i0 = (m - b)*D, tap k = b*D - u.  The u axis [u_min, u_min + 4S), u_min = -(nt-1),
 * S = ceil((nt + 15 D)/4) rounded up to a multiple of 4, is cut into 4 equal quarters; quarter g is
 * ONE fmaf chain, u ascending (oldest sample first), starting from +0;
 * y = (r0 + r1) + (r2 + r3).  Taps outside [0, nt) and samples outside the stream are +0 and DO take
 * part (same value as skipping them, but it fixes the sign of a zero result; inputs must be finite). */
int orc_m16_steps(int nt, int D)
{
    int S = (nt + 15 * D + 3) / 4;
    return (S + 3) / 4 * 4;
}
size_t orc_decim_fir_ccf_m16(const cf32* in, size_t n, const float* taps, int nt, int D, cf32* out)
{
    size_t nout = orc_decim_count(n, 1, D);
    const int S = orc_m16_steps(nt, D), Sq = S / 4;
    const long long u_min = -(long long)(nt - 1);
    for (size_t m = 0; m < nout; m++) {
        const int b = (int)(m & 15u);
        const long long i0 = (long long)(m - (size_t)b) * D;
        float rr[4], ri[4];
        for (int g = 0; g < 4; g++) {
            float ar = 0.0f, ai = 0.0f;
            const long long ua = u_min + 4LL * Sq * g, ub = ua + 4LL * Sq;
            /* every u of the quarter takes part, exactly as in the matrix product: taps outside the band
             * are +0 and samples outside the stream are +0.  (Skipping them would be the same value except
             * for the SIGN OF ZERO: fmaf(0, x, -0) is +0 when 0*x is +0.) */
            for (long long u = ua; u < ub; u++) {
                const long long k = (long long)b * D - u, i = i0 + u;
                const float h = (k >= 0 && k < nt) ? taps[k] : 0.0f;
                cf32 x = {0.0f, 0.0f};
                if (i >= 0 && (size_t)i < n) x = in[i];
                ar = fmaf(h, x.re, ar);
                ai = fmaf(h, x.im, ai);
            }
            rr[g] = ar; ri[g] = ai;
        }
        out[m].re = (rr[0] + rr[1]) + (rr[2] + rr[3]);
        out[m].im = (ri[0] + ri[1]) + (ri[2] + ri[3]);
    }
    return nout;
}
/* Which contract a (nt, D) decimator uses; libqrl_hip's DecimStage::plan applies the same rule:
 * m16 when D >= 8 and the smallest MFMA tile (8 output blocks of 16) fits the LDS budget. */
int orc_decim_uses_m16(int nt, int D)
{
    if (D < 8) return 0;
    const long long samples = 7LL * 16 * D + 4LL * orc_m16_steps(nt, D);
    return (samples + 2 * (samples / (16LL * D)) + 64) * 8 <= 150 * 1024;
}
/* "pl" (phase-lane) summation contract of the register-resident decimator k_decim_pl.
 * Samples are dealt to 64 lane slots: super-block length D' = R*D with R = 64/D (integer division) when
 * D <= 32, else D' = D; a slot holds E = ceil(D'/64) consecutive samples; slot(i) = ((i - 1) mod D') / E
 * (block c covers samples (c-1)D'+1 .. cD').  For output m every slot forms ONE chain over its own samples
 * i (ascending = oldest first) with tap k = m*D - i in [0, nt): the first term is the plain product
 * h*x, every later term an fmaf; a slot without samples holds +0.  Samples before the stream start are +0
 * and take part.  The 64 slot values meet in a radix-2 tree: for h = 32,16,8,4,2,1: v[l] = v[l] + v[l+h], l < h.
 * (k_decim_pl also runs the zero taps k >= nt of a slot's chain; that can only change the sign of an exact zero.) */
void orc_pl_geometry(int D, int* Dp, int* E)
{
    const int R = D <= 32 ? 64 / D : 1;
    *Dp = R * D;
    *E = (*Dp + 63) / 64;
}
int orc_decim_uses_pl(int nt, int D)
{
    /* the geometries the register-resident kernels are instantiated for (qradiolink_amd/csrc/kernels_decim_pl.hip pl_geom):
     * one sample per lane and block with at most 16 taps per lane (32 < D <= 64), or two samples per lane (64 < D <= 128, D even)
     * with at most 42 outputs in flight per sample */
    int Dp, E;
    orc_pl_geometry(D, &Dp, &E);
    const int R = Dp / D, U = (nt + Dp - 1) / D;
    if (E == 1 && R == 1) return 0;           /* 32 < D <= 64: the phase-major matrix-pipe contract "pm" below */
    if (E == 2 && R == 1) return (D % 2) == 0 && U <= 42;
    return 0;
}
/* ---- contract "pm" (phase major; qradiolink_amd/csrc/kernels_decim_pl.hip k_decim_pm) ---------------------------------------------
 * The stream is cut into BLOCKS of D samples, block c = samples (c-1) D + 1 .. c D (so output m ends at the last sample of block m).
 * Sample p (0 .. D-1) of block c = m - j meets output m with tap H(p, j) = h[j D + D - 1 - p], j = 0 .. J-1, J = ceil(nt / D) <= 48.
 * Lags come in TILES of 16, j = 16 t + j' (NT = ceil(J / 16) <= 3 tiles).  For a block c and a lag-in-tile j':
 *   Z(c, j') = ONE fmaf chain from +0 over the tiles t = NT-1 .. 0 (the oldest data first) and, inside a tile, the D samples of block
 *              c - 16 t with p ASCENDING -- the term of output c + j' that the lags j', 16 + j', 32 + j' contribute
 *              (v_mfma_f32_16x16x4_f32 accumulates k in order; the accumulator of an output group walks through three consecutive
 *              groups of blocks, one tile at a time, before it is read).
 * One matrix result holds Z for the 16 blocks n of an ABSOLUTE group (blocks 16 G .. 16 G + 15) and the 16 lags j', lag j' = 4 q + r in
 * lane row q, register r.  The 16 terms of an output are then added with plain float adds:
 *   L_q[n]  = ((Z(n, 4q) + Z(n-1, 4q+1)) + Z(n-2, 4q+2)) + Z(n-3, 4q+3)          blocks of the same group, missing ones (n - r < 0) = +0
 *   H_q[k]  = (Z(15+k, 4q+1) + Z(14+k, 4q+2)) + Z(13+k, 4q+3)                    k = 0 .. 2: what leaves the group, missing ones = +0
 *   output 16 G + n':   R_q = L_q(G)[n' - 4q]               if n' >= 4q, else +0          (in-row terms)
 *                       C_q = L_q(G-1)[n' + 16 - 4q]        if n' <  4q, else H_q(G-1)[n' - 4q]   (+0 for n' - 4q > 2)   (carries)
 *                       y = ((R_0 + C_0) + (R_1 + C_1)) + ((R_2 + C_2) + (R_3 + C_3))
 * For one tile this is the order of rounds-3a's first kernel (terms of a lane row j' ascending, in-row and carried terms separately).
 * The grouping is by absolute index, so the value of output m does not depend on where a call or a kernel segment starts.
 * Geometries: the 1:50-class first stages (32 < D <= 52, J <= 16: one tile) and the device-rate front ends of 10 / 20 / 25 / 50 /
 * 100 Msps (41.8 D taps: J = 42, three tiles), i.e. ceil(D / 4) in {3, 5, 7, 13, 25}: the instantiated kernels. */
int orc_decim_uses_pm(int nt, int D)
{
    const int J = (nt + D - 1) / D, NS = (D + 3) / 4;
    if (D > 32 && D <= 52 && J <= 16) return 1;
    return J > 16 && J <= 48 && (NS == 3 || NS == 5 || NS == 7 || NS == 13 || NS == 25);
}
typedef struct { const cf32* in; size_t n; const float* taps; int nt, D, NT; } pm_ctx;
/* Z(c, j') of the contract; c may be negative or beyond the data (zero samples) */
static void pm_z(const pm_ctx* k, long long c, int jp, float* zr, float* zi)
{
    float ar = 0.0f, ai = 0.0f;
    for (int t = k->NT - 1; t >= 0; t--) {
        const int j = 16 * t + jp;
        const long long cb = c - 16LL * t;
        for (int p = 0; p < k->D; p++) {
            const long long kk = (long long)j * k->D + k->D - 1 - p;
            const long long i = (cb - 1) * (long long)k->D + 1 + p;
            const float h = kk < k->nt ? k->taps[kk] : 0.0f;
            cf32 x = {0.0f, 0.0f};
            if (i >= 0 && (size_t)i < k->n) x = k->in[i];
            ar = fmaf(h, x.re, ar); ai = fmaf(h, x.im, ai);
        }
    }
    *zr = ar; *zi = ai;
}
static void pm_L(const pm_ctx* k, long long G, int q, int n, float* lr, float* li)
{
    float ar = 0.0f, ai = 0.0f;
    for (int r = 0; r < 4; r++) {
        float zr = 0.0f, zi = 0.0f;
        if (n - r >= 0) pm_z(k, 16 * G + n - r, 4 * q + r, &zr, &zi);
        if (r == 0) { ar = zr; ai = zi; } else { ar = ar + zr; ai = ai + zi; }
    }
    *lr = ar; *li = ai;
}
static void pm_H(const pm_ctx* k, long long G, int q, int kq, float* hr, float* hi)
{
    float ar = 0.0f, ai = 0.0f;
    for (int r = 1; r < 4; r++) {
        float zr = 0.0f, zi = 0.0f;
        if (kq <= r - 1) pm_z(k, 16 * G + 16 - r + kq, 4 * q + r, &zr, &zi);
        if (r == 1) { ar = zr; ai = zi; } else { ar = ar + zr; ai = ai + zi; }
    }
    *hr = ar; *hi = ai;
}
size_t orc_decim_fir_ccf_pm(const cf32* in, size_t n, const float* taps, int nt, int D, cf32* out)
{
    const size_t nout = orc_decim_count(n, 1, D);
    const int J = (nt + D - 1) / D;
    const pm_ctx k = {in, n, taps, nt, D, (J + 15) / 16};
    for (size_t m = 0; m < nout; m++) {
        const long long G = (long long)(m >> 4);
        const int np = (int)(m & 15u);
        float Vr[4], Vi[4];
        for (int q = 0; q < 4; q++) {
            float Rr = 0.0f, Ri = 0.0f, Cr = 0.0f, Ci = 0.0f;
            if (np >= 4 * q) pm_L(&k, G, q, np - 4 * q, &Rr, &Ri);
            if (np < 4 * q) pm_L(&k, G - 1, q, np + 16 - 4 * q, &Cr, &Ci);
            else if (np - 4 * q <= 2) pm_H(&k, G - 1, q, np - 4 * q, &Cr, &Ci);
            Vr[q] = Rr + Cr; Vi[q] = Ri + Ci;
        }
        out[m].re = (Vr[0] + Vr[1]) + (Vr[2] + Vr[3]);
        out[m].im = (Vi[0] + Vi[1]) + (Vi[2] + Vi[3]);
    }
    return nout;
}
size_t orc_decim_fir_ccf_pl(const cf32* in, size_t n, const float* taps, int nt, int D, cf32* out)
{
    size_t nout = orc_decim_count(n, 1, D);
    int Dp, E;
    orc_pl_geometry(D, &Dp, &E);
    for (size_t m = 0; m < nout; m++) {
        float vr[64], vi[64];
        int used[64];
        for (int l = 0; l < 64; l++) { vr[l] = 0.0f; vi[l] = 0.0f; used[l] = 0; }
        for (int k = nt - 1; k >= 0; k--) {
            const long long i = (long long)m * D - k;
            cf32 x = {0.0f, 0.0f};
            if (i >= 0) x = in[i];
            long long r = (i - 1) % Dp;
            if (r < 0) r += Dp;
            const int l = (int)(r / E);
            const float h = taps[k];
            if (!used[l]) { vr[l] = h * x.re; vi[l] = h * x.im; used[l] = 1; }
            else { vr[l] = fmaf(h, x.re, vr[l]); vi[l] = fmaf(h, x.im, vi[l]); }
        }
        for (int h = 32; h >= 1; h >>= 1)
            for (int l = 0; l < h; l++) { vr[l] = vr[l] + vr[l + h]; vi[l] = vi[l] + vi[l + h]; }
        out[m].re = vr[0]; out[m].im = vi[0];
    }
    return nout;
}
/* CPU-BASELINE variant of the decimating FIR (bench.py cpu_baseline leg only, never the checker): the dot product as a
 * VOLK-like AVX2 kernel (volk_32fc_32f_dot_prod_32fc: complex samples x real taps, 8 floats per fused multiply-add, four
 * independent accumulators, horizontal sum at the end).  Same value as the contracts above to ~1e-6 of RMS, different
 * rounding order, so it is selected explicitly with orc_set_decim_impl(1) and only timed. */
#include <immintrin.h>
static int g_decim_impl = 0;
void orc_set_decim_impl(int impl) { g_decim_impl = impl; }
size_t orc_decim_fir_ccf_simd(const cf32* in, size_t n, const float* taps, int nt, int D, cf32* out)
{
    size_t nout = orc_decim_count(n, 1, D);
    /* reversed taps, each duplicated for (re, im), zero padded to a multiple of 32 floats */
    const int nf = 2 * nt, nfp = (nf + 31) / 32 * 32;
    float* hr = (float*)aligned_alloc(32, (size_t)nfp * sizeof(float));
    for (int i = 0; i < nfp; i++) hr[i] = 0.0f;
    for (int k = 0; k < nt; k++) { hr[2 * (nt - 1 - k)] = taps[k]; hr[2 * (nt - 1 - k) + 1] = taps[k]; }
    const float* xf = (const float*)in;
    for (size_t m = 0; m < nout; m++) {
        const long long first = (long long)m * D - (nt - 1);        /* oldest sample of the window */
        if (first < 0 || (size_t)(first + nfp / 2) > n) {           /* stream edges: plain loop */
            float ar = 0.0f, ai = 0.0f;
            for (int k = 0; k < nt; k++) {
                const long long i = (long long)m * D - k;
                if (i < 0) break;
                ar = fmaf(taps[k], in[i].re, ar); ai = fmaf(taps[k], in[i].im, ai);
            }
            out[m].re = ar; out[m].im = ai;
            continue;
        }
        const float* x = xf + 2 * first;
        __m256 a0 = _mm256_setzero_ps(), a1 = a0, a2 = a0, a3 = a0;
        for (int j = 0; j < nfp; j += 32) {
            a0 = _mm256_fmadd_ps(_mm256_load_ps(hr + j), _mm256_loadu_ps(x + j), a0);
            a1 = _mm256_fmadd_ps(_mm256_load_ps(hr + j + 8), _mm256_loadu_ps(x + j + 8), a1);
            a2 = _mm256_fmadd_ps(_mm256_load_ps(hr + j + 16), _mm256_loadu_ps(x + j + 16), a2);
            a3 = _mm256_fmadd_ps(_mm256_load_ps(hr + j + 24), _mm256_loadu_ps(x + j + 24), a3);
        }
        float v[8];
        _mm256_storeu_ps(v, _mm256_add_ps(_mm256_add_ps(a0, a1), _mm256_add_ps(a2, a3)));
        out[m].re = (v[0] + v[2]) + (v[4] + v[6]);
        out[m].im = (v[1] + v[3]) + (v[5] + v[7]);
    }
    free(hr);
    return nout;
}
/* the decimator of the frequency-translating multi-carrier graphs: the product keeps the banded-Toeplitz MFMA kernel there (chan.cpp),
 * so the summation order is the "m16" contract whatever orc_decim_auto's rule says for the geometry */
size_t orc_decim_xlating(const cf32* in, size_t n, const float* taps, int nt, int D, cf32* out)
{
    orc_trace_event("resamp_ccf(1,%d,%s)", D, orc_trace_name(taps, sizeof(float) * (size_t)nt));
    if (g_decim_impl == 1) return orc_decim_fir_ccf_simd(in, n, taps, nt, D, out);
    if (orc_decim_uses_m16(nt, D)) return orc_decim_fir_ccf_m16(in, n, taps, nt, D, out);
    return orc_decim_fir_ccf(in, n, taps, nt, D, 4, out);
}
size_t orc_decim_auto(const cf32* in, size_t n, const float* taps, int nt, int D, cf32* out)
{
    orc_trace_event("resamp_ccf(1,%d,%s)", D, orc_trace_name(taps, sizeof(float) * (size_t)nt));
    if (g_decim_impl == 1) return orc_decim_fir_ccf_simd(in, n, taps, nt, D, out);
    if (orc_decim_uses_pm(nt, D)) return orc_decim_fir_ccf_pm(in, n, taps, nt, D, out);
    if (orc_decim_uses_pl(nt, D)) return orc_decim_fir_ccf_pl(in, n, taps, nt, D, out);
    if (orc_decim_uses_m16(nt, D)) return orc_decim_fir_ccf_m16(in, n, taps, nt, D, out);
    return orc_decim_fir_ccf(in, n, taps, nt, D, 4, out);
}

/* General I/D.  One fmaf chain per output, j ascending. */
size_t orc_resamp_ccf(const cf32* in, size_t n, const float* taps, int nt, int I, int D, cf32* out)
{
    orc_trace_event("resamp_ccf(%d,%d,%s)", I, D, orc_trace_name(taps, sizeof(float) * (size_t)nt));
    size_t nout = orc_decim_count(n, I, D);
    for (size_t i = 0; i < nout; i++) {
        uint64_t u = (uint64_t)i * (uint64_t)D;
        int ph = (int)(u % (uint64_t)I);
        long long c = (long long)(u / (uint64_t)I);
        float ar = 0.0f, ai = 0.0f;
        for (int j = 0; ph + j * I < nt; j++) {
            long long idx = c - j;
            if (idx < 0) break;
            float h = taps[ph + j * I];
            ar = fmaf(h, in[idx].re, ar);
            ai = fmaf(h, in[idx].im, ai);
        }
        out[i].re = ar; out[i].im = ai;
    }
    return nout;
}
size_t orc_resamp_fff(const float* in, size_t n, const float* taps, int nt, int I, int D, float* out)
{
    orc_trace_event("resamp_fff(%d,%d,%s)", I, D, orc_trace_name(taps, sizeof(float) * (size_t)nt));
    size_t nout = orc_decim_count(n, I, D);
    for (size_t i = 0; i < nout; i++) {
        uint64_t u = (uint64_t)i * (uint64_t)D;
        int ph = (int)(u % (uint64_t)I);
        long long c = (long long)(u / (uint64_t)I);
        float a = 0.0f;
        for (int j = 0; ph + j * I < nt; j++) {
            long long idx = c - j;
            if (idx < 0) break;
            a = fmaf(taps[ph + j * I], in[idx], a);
        }
        out[i] = a;
    }
    return nout;
}

/* ------------------------------------------------------------------------------------------
 * fft_filter_{ccf,ccc,fff} restated as the causal direct FIR they implement
 * [gr-filter/lib/fft_filter.cc]: y[n] = sum_k h[k] x[n-k], one fmaf chain, k ascending.
 * ------------------------------------------------------------------------------------------ */
void orc_fir_ccf(const cf32* in, size_t n, const float* taps, int nt, cf32* out)
{
    orc_trace_event("fir_ccf(%s)", orc_trace_name(taps, sizeof(float) * (size_t)nt));
    for (size_t i = 0; i < n; i++) {
        float ar = 0.0f, ai = 0.0f;
        for (int k = 0; k < nt && (size_t)k <= i; k++) {
            ar = fmaf(taps[k], in[i - k].re, ar);
            ai = fmaf(taps[k], in[i - k].im, ai);
        }
        out[i].re = ar; out[i].im = ai;
    }
}
/* complex taps: re += hr*xr; re += (-hi)*xi; im += hr*xi; im += hi*xr */
void orc_fir_ccc(const cf32* in, size_t n, const cf32* taps, int nt, cf32* out)
{
    orc_trace_event("fir_ccc(%s)", orc_trace_name(taps, sizeof(cf32) * (size_t)nt));
    for (size_t i = 0; i < n; i++) {
        float ar = 0.0f, ai = 0.0f;
        for (int k = 0; k < nt && (size_t)k <= i; k++) {
            cf32 h = taps[k], x = in[i - k];
            ar = fmaf(h.re, x.re, ar);
            ar = fmaf(-h.im, x.im, ar);
            ai = fmaf(h.re, x.im, ai);
            ai = fmaf(h.im, x.re, ai);
        }
        out[i].re = ar; out[i].im = ai;
    }
}
/* The upper / lower band-edge discriminator filters of gr_demod_2fsk (complex_band_pass(-W, 0) and (0, W), gr_demod_2fsk.cpp:86-89)
 * are a CONJUGATE pair, bit for bit (same low-pass prototype, cos(-x) = cos(x), sin(-x) = -sin(x)).  With h = a + j b:
 *   A = sum a[k] x[i-k]   (real taps on the complex samples: two fmaf chains, k ascending)
 *   B = sum b[k] x[i-k]
 *   upper = (A.re - B.im, A.im + B.re),   lower = (A.re + B.im, A.im - B.re)
 * -- half the multiplies of two independent complex-tap filters.  This is the summation contract of the product for such a pair
 * (kernels_ff.hip k_2fsk_ff / k_disc_2fsk); a pair that is not conjugate falls back to two orc_fir_ccc. */
void orc_fir_ccc_conj_pair(const cf32* in, size_t n, const cf32* up, const cf32* lo, int nt, cf32* out_up, cf32* out_lo)
{
    int conj = 1;
    for (int k = 0; k < nt; k++) {
        float nb = -up[k].im;
        if (memcmp(&lo[k].re, &up[k].re, sizeof(float)) != 0 || memcmp(&lo[k].im, &nb, sizeof(float)) != 0) { conj = 0; break; }
    }
    if (!conj) { orc_fir_ccc(in, n, up, nt, out_up); orc_fir_ccc(in, n, lo, nt, out_lo); return; }
    orc_trace_event("fir_ccc(%s)", orc_trace_name(up, sizeof(cf32) * (size_t)nt));
    orc_trace_event("fir_ccc(%s)", orc_trace_name(lo, sizeof(cf32) * (size_t)nt));
    for (size_t i = 0; i < n; i++) {
        float ar = 0.0f, ai = 0.0f, br = 0.0f, bi = 0.0f;
        for (int k = 0; k < nt && (size_t)k <= i; k++) {
            ar = fmaf(up[k].re, in[i - k].re, ar); ai = fmaf(up[k].re, in[i - k].im, ai);
            br = fmaf(up[k].im, in[i - k].re, br); bi = fmaf(up[k].im, in[i - k].im, bi);
        }
        out_up[i].re = ar - bi; out_up[i].im = ai + br;
        out_lo[i].re = ar + bi; out_lo[i].im = ai - br;
    }
}
void orc_fir_fff(const float* in, size_t n, const float* taps, int nt, float* out)
{
    orc_trace_event("fir_fff(%s)", orc_trace_name(taps, sizeof(float) * (size_t)nt));
    for (size_t i = 0; i < n; i++) {
        float a = 0.0f;
        for (int k = 0; k < nt && (size_t)k <= i; k++) a = fmaf(taps[k], in[i - k], a);
        out[i] = a;
    }
}

/* ------------------------------------------------------------------------------------------
 * control_loop helpers [gr-blocks control_loop.h]
 * ------------------------------------------------------------------------------------------ */
static inline float phase_wrap(float phase)
{
    while (phase > (float)(2 * M_PI))  phase = (float)((double)phase - 2 * M_PI);
    while (phase < (float)(-2 * M_PI)) phase = (float)((double)phase + 2 * M_PI);
    return phase;
}
static inline float branchless_clip(float x, float clip)
{
    /* gr::branchless_clip: 0.5*(|x+clip| - |x-clip|) */
    float x1 = fabsf(x + clip);
    float x2 = fabsf(x - clip);
    x1 -= x2;
    return 0.5f * x1;
}

/* ------------------------------------------------------------------------------------------
 * fll_band_edge_cc [gr-digital/lib/fll_band_edge_cc_impl.cc]
 * history = ntaps+1 and out[i] = in[i]*nco  =>  the stream is delayed by ntaps samples.
 * The two band-edge FIRs run on the derotated output; filter taps as stored by design_filter
 * (reversed) and applied by fir_filter_with_buffer_ccc: f = sum_j T[j] * y[n-j] with
 * T[j] = stored[ntaps-1-j].  Complex MAC order as orc_fir_ccc.
 * ------------------------------------------------------------------------------------------ */
void orc_fll_band_edge(const cf32* in, size_t n, float sps, float rolloff, int nt, float bw, cf32* out)
{
    orc_trace_event("fll_band_edge(%.9g,%.9g,%d,%.9g)", sps, rolloff, nt, bw);
    cf32* lower = (cf32*)malloc(sizeof(cf32) * (size_t)nt);
    cf32* upper = (cf32*)malloc(sizeof(cf32) * (size_t)nt);
    cf32* dl = (cf32*)calloc((size_t)nt, sizeof(cf32)); /* dl[j] = y[n-j] */
    orc_fll_taps(sps, rolloff, nt, lower, upper);
    float alpha, beta;
    orc_control_loop_gains(bw, &alpha, &beta);
    float max_freq = (float)(2 * M_PI * (2.0 / sps)), min_freq = -max_freq;
    float phase = 0, freq = 0;
    for (size_t i = 0; i < n; i++) {
        cf32 x = {0, 0};
        if (i >= (size_t)nt) x = in[i - (size_t)nt];
        cf32 nco; orc_sincosf(phase, &nco.im, &nco.re);
        cf32 y;
        y.re = x.re * nco.re - x.im * nco.im;
        y.im = x.re * nco.im + x.im * nco.re;
        out[i] = y;
        memmove(dl + 1, dl, sizeof(cf32) * (size_t)(nt - 1));
        dl[0] = y;
        /* Summation contract (what the 4-lanes-per-stream GPU kernel does): the delay line is cut into 4 groups of nt/4
         * consecutive samples, group g = dl[g*nt/4 .. (g+1)*nt/4 - 1]; inside a group one fmaf chain per accumulator, OLDEST
         * sample first (only the last link of group 0 depends on y[n]); the partial sums meet as (p0 + p1) + (p2 + p3). */
        float pur[4], pui[4], plr[4], pli[4];
        const int gl = nt / 4;
        for (int g = 0; g < 4; g++) {
            float ur = 0, ui = 0, lr = 0, li = 0;
            for (int j = (g + 1) * gl - 1; j >= g * gl; j--) {
                cf32 hu = upper[nt - 1 - j], hl = lower[nt - 1 - j], v = dl[j];
                ur = fmaf(hu.re, v.re, ur); ur = fmaf(-hu.im, v.im, ur);
                ui = fmaf(hu.re, v.im, ui); ui = fmaf(hu.im, v.re, ui);
                lr = fmaf(hl.re, v.re, lr); lr = fmaf(-hl.im, v.im, lr);
                li = fmaf(hl.re, v.im, li); li = fmaf(hl.im, v.re, li);
            }
            pur[g] = ur; pui[g] = ui; plr[g] = lr; pli[g] = li;
        }
        const float ur = (pur[0] + pur[1]) + (pur[2] + pur[3]), ui = (pui[0] + pui[1]) + (pui[2] + pui[3]);
        const float lr = (plr[0] + plr[1]) + (plr[2] + plr[3]), li = (pli[0] + pli[1]) + (pli[2] + pli[3]);
        float error = (lr * lr + li * li) - (ur * ur + ui * ui);
        freq = freq + beta * error;
        phase = phase + freq + alpha * error;
        phase = phase_wrap(phase);
        if (freq > max_freq) freq = max_freq; else if (freq < min_freq) freq = min_freq;
    }
    free(lower); free(upper); free(dl);
}

/* quadrature_demod_cf [gr-analog/lib/quadrature_demod_cf_impl.cc]: history 2 */
void orc_quad_demod(const cf32* in, size_t n, float gain, float* out)
{
    orc_trace_event("quad_demod(%.9g)", gain);
    cf32 prev = {0, 0};
    for (size_t i = 0; i < n; i++) {
        cf32 a = in[i];
        float re = a.re * prev.re + a.im * prev.im;
        float im = a.im * prev.re - a.re * prev.im;
        out[i] = gain * orc_fast_atan2f(im, re);
        prev = a;
    }
}

/* agc2_cc [gr-analog include/gnuradio/analog/agc2.h] */
void orc_agc2(const cf32* in, size_t n, float attack, float decay, float ref, float gain, float max_gain, cf32* out)
{
    orc_trace_event("agc2_cc(%.9g,%.9g,%.9g,%.9g,%.9g)", attack, decay, ref, gain, max_gain);
    for (size_t i = 0; i < n; i++) {
        cf32 o; o.re = in[i].re * gain; o.im = in[i].im * gain;
        float tmp = -ref + sqrtf(o.re * o.re + o.im * o.im);
        float rate = decay;
        if (tmp > gain) rate = attack;
        gain -= tmp * rate;
        if (gain < 0.0f) gain = 10e-5f;
        if (max_gain > 0.0f && gain > max_gain) gain = max_gain;
        out[i] = o;
    }
}

/* costas_loop_cc [gr-digital/lib/costas_loop_cc_impl.cc] */
void orc_costas(const cf32* in, size_t n, float bw, int order, int use_snr, cf32* out)
{
    orc_trace_event("costas(%.9g,%d,%d)", bw, order, use_snr);
    float alpha, beta;
    orc_control_loop_gains(bw, &alpha, &beta);
    float phase = 0, freq = 0;
    for (size_t i = 0; i < n; i++) {
        cf32 nco; orc_sincosf(-phase, &nco.im, &nco.re);
        cf32 x = in[i], o;
        o.re = x.re * nco.re - x.im * nco.im;
        o.im = x.re * nco.im + x.im * nco.re;
        out[i] = o;
        float e;
        if (order == 2) {
            if (use_snr) { float snr = (o.re * o.re + o.im * o.im); e = orc_tanhf_lut(snr * o.re) * o.im; }
            else e = o.re * o.im;
        } else {
            if (use_snr) {
                float snr = (o.re * o.re + o.im * o.im);
                e = (orc_tanhf_lut(snr * o.re) * o.im) - (orc_tanhf_lut(snr * o.im) * o.re);
            } else {
                e = ((o.re > 0 ? 1.0f : -1.0f) * o.im - (o.im > 0 ? 1.0f : -1.0f) * o.re);
            }
        }
        e = branchless_clip(e, 1.0f);
        freq = freq + beta * e;
        phase = phase + freq + alpha * e;
        phase = phase_wrap(phase);
        if (freq > 1.0f) freq = 1.0f; else if (freq < -1.0f) freq = -1.0f;
    }
}

/* ------------------------------------------------------------------------------------------
 * symbol_sync_{ff,cc} [gr-digital/lib/symbol_sync_ff_impl.cc, timing_error_detector.cc,
 * clock_tracking_loop.cc, interpolating_resampler.cc, gr-filter mmse_fir_interpolator_*]
 * osps = 1, IR_MMSE_8TAP.  A symbol at cursor ii exists iff in[ii..ii+7] exist.
 * interp(in, mu) = sum_{k=0..7} T[imu][7-k] * in[k], imu = rint(mu*128), one fmaf chain k ascending.
 * ------------------------------------------------------------------------------------------ */
/* Modified M&M TED: the formula is a NAMED contract (include/qrl_contracts.h) shared with the HIP kernels.  The override exists
 * for tests/test_ted_sensitivity.py only: it measures which hard decisions each candidate formula would move. */
static int g_ted_ff = QRL_TED_MODMM_FF, g_ted_cc = QRL_TED_MODMM_CC;
void orc_set_ted_modmm(int ff_variant, int cc_variant)
{
    g_ted_ff = ff_variant < 0 ? QRL_TED_MODMM_FF : ff_variant;
    g_ted_cc = cc_variant < 0 ? QRL_TED_MODMM_CC : cc_variant;
}
void orc_get_ted_modmm(int* ff_variant, int* cc_variant) { *ff_variant = g_ted_ff; *cc_variant = g_ted_cc; }

static inline float slice_real(int constellation, float x)
{
    if (constellation == ORC_CONST_BPSK) return x > 0 ? 1.0f : -1.0f;
    /* constellation_rect::make(points {-1.5,-0.5,0.5,1.5}, -, 2, 4, 1, 1.0, 1.0) [gr-digital constellation.cc:
     * constellation_sector::decision_maker -> constellation_rect::get_sector]: real sector =
     * (int)(re / width_real + n_real / 2.0) clamped to [0, 3] (boundaries at -1, 0, 1; a value ON a boundary, e.g. the
     * zeros before the signal starts, goes to the upper sector); the sector's point is the one closest to its centre. */
    int sector = (int)((double)x / 1.0 + 2.0);
    if (sector < 0) sector = 0; else if (sector > 3) sector = 3;
    return (float)sector - 1.5f;
}

/* how often a timing loop sat in its limiter since the last reset (tests only: tests/test_channel_8d.py shows that the clock-error
 * cases of the parity lists really drive symbol_sync's max_dev clamp / clock_recovery_mm's omega limit; not thread safe, read it
 * around single-threaded chain calls) */
static size_t g_clamp_hits = 0;
size_t orc_loop_clamp_hits(int reset) { size_t v = g_clamp_hits; if (reset) g_clamp_hits = 0; return v; }

typedef struct { float avg, inst, alpha, beta, maxp, minp; } clock_loop;
static inline void clock_advance(clock_loop* c, float e)
{
    c->avg = c->avg + c->beta * e;
    if (c->avg > c->maxp) { c->avg = c->maxp; g_clamp_hits++; } else if (c->avg < c->minp) { c->avg = c->minp; g_clamp_hits++; }
    c->inst = c->avg + c->alpha * e;
    if (c->inst <= 0.f) c->inst = c->avg;
}

size_t orc_symbol_sync_ff(const float* in, size_t n, int ted, float sps, float loop_bw, float damping,
                          float ted_gain, float max_dev, int constellation, float* out)
{
    orc_trace_event("symbol_sync_ff(%d,%.9g,%.9g,%.9g,%.9g,%.9g,%d)", ted, sps, loop_bw, damping, ted_gain, max_dev, constellation);
    const float* T = orc_mmse_table();
    clock_loop c; c.avg = sps; c.inst = sps; c.maxp = sps + max_dev; c.minp = sps - max_dev;
    orc_clock_loop_gains(loop_bw, damping, ted_gain, &c.alpha, &c.beta);
    float x0 = 0, x1 = 0, x2 = 0, d0 = 0, d1 = 0, d2 = 0;
    float mu = 0; size_t ii = 0, oo = 0;
    while (ii + 8 <= n) {
        int imu = (int)rintf(mu * 128.0f);
        const float* t = T + imu * 8;
        float y = 0.0f;
        for (int k = 0; k < 8; k++) y = fmaf(t[7 - k], in[ii + (size_t)k], y);
        x2 = x1; x1 = x0; x0 = y;
        d2 = d1; d1 = d0; d0 = slice_real(constellation, y);
        float e;
        if (ted == ORC_TED_MM) e = d1 * x0 - d0 * x1;
        else { float u = ((x0 - x2) * d1) - ((d0 - d2) * x1); e = QRL_TED_MODMM_ERROR(g_ted_ff, u, branchless_clip); }
        clock_advance(&c, e);
        float phase = mu + c.inst;
        float fl = floorf(phase);
        mu = phase - fl;
        out[oo++] = y;
        ii += (size_t)(int)fl;
    }
    return oo;
}

size_t orc_symbol_sync_cc(const cf32* in, size_t n, int ted, float sps, float loop_bw, float damping,
                          float ted_gain, float max_dev, int constellation, cf32* out)
{
    orc_trace_event("symbol_sync_cc(%d,%.9g,%.9g,%.9g,%.9g,%.9g,%d)", ted, sps, loop_bw, damping, ted_gain, max_dev, constellation);
    const float* T = orc_mmse_table();
    /* dqpsk: (+-0.707107, +-0.707107) by sign; 4-level rect (gr_demod_4fsk non-FM branch): real-axis sector point, imag 0 */
    clock_loop c; c.avg = sps; c.inst = sps; c.maxp = sps + max_dev; c.minp = sps - max_dev;
    orc_clock_loop_gains(loop_bw, damping, ted_gain, &c.alpha, &c.beta);
    cf32 x0 = {0, 0}, x1 = {0, 0}, x2 = {0, 0}, d0 = {0, 0}, d1 = {0, 0}, d2 = {0, 0};
    float mu = 0; size_t ii = 0, oo = 0;
    const float SQ = 0.707107f;
    while (ii + 8 <= n) {
        int imu = (int)rintf(mu * 128.0f);
        const float* t = T + imu * 8;
        cf32 y = {0.0f, 0.0f};
        for (int k = 0; k < 8; k++) {
            y.re = fmaf(t[7 - k], in[ii + (size_t)k].re, y.re);
            y.im = fmaf(t[7 - k], in[ii + (size_t)k].im, y.im);
        }
        x2 = x1; x1 = x0; x0 = y;
        d2 = d1; d1 = d0;
        if (constellation == ORC_CONST_4LEVEL) { d0.re = slice_real(ORC_CONST_4LEVEL, y.re); d0.im = 0.0f; }
        else { d0.re = y.re > 0 ? SQ : -SQ; d0.im = y.im > 0 ? SQ : -SQ; }
        float e;
        if (ted == ORC_TED_MM) {
            e = (d1.re * x0.re - d0.re * x1.re) + (d1.im * x0.im - d0.im * x1.im);
        } else {
            float ar = x0.re - x2.re, ai = x0.im - x2.im;
            float br = d0.re - d2.re, bi = d0.im - d2.im;
            float u = (ar * d1.re + ai * d1.im) - (br * x1.re + bi * x1.im);
            e = QRL_TED_MODMM_ERROR(g_ted_cc, u, branchless_clip);
        }
        clock_advance(&c, e);
        float phase = mu + c.inst;
        float fl = floorf(phase);
        mu = phase - fl;
        out[oo++] = y;
        ii += (size_t)(int)fl;
    }
    return oo;
}

void orc_diff_phasor(const cf32* in, size_t n, cf32* out)
{
    orc_trace_event("diff_phasor()");
    cf32 prev = {0, 0};
    for (size_t i = 0; i < n; i++) {
        cf32 a = in[i];
        out[i].re = a.re * prev.re + a.im * prev.im;
        out[i].im = a.im * prev.re - a.re * prev.im;
        prev = a;
    }
}

/* multiply_const_ff -> add_const_ff -> float_to_uchar: sat_0^255(rint(x*mul + add)) */
void orc_soft_quant(const float* in, size_t n, float mul, float add, uint8_t* out)
{
    orc_trace_event("soft_quant(%.9g,%.9g)", mul, add);
    for (size_t i = 0; i < n; i++) {
        float v = in[i] * mul;
        v = v + add;
        float r = rintf(v);
        if (!(r >= 0.0f)) r = 0.0f;          /* NaN -> 0 */
        if (r > 255.0f) r = 255.0f;
        out[i] = (uint8_t)r;
    }
}

/* ------------------------------------------------------------------------------------------
 * fec::decoder(cc_decoder(frame 80, K 7, rate 2, polys {109,79}, CC_STREAMING))
 * [gr-fec/lib/cc_decoder_impl.cc, decoder_impl.cc; VOLK volk_8u_x4_conv_k7_r2_8u_spiral]
 * block b exists iff soft[160b .. 160b+171] exist (160 consumed + 12 look-ahead).
 * The metric kernel restated is the SPIRAL variant -- the one the reference says decodes
 * correctly (docs/OPERATION.md:4): branch metric = (avg_epu8(B0^s0, B1^s1) >> 2) & 63, all
 * path-metric adds SATURATE at 255 (_mm_adds_epu8), survivor = min with decision bit 1 on
 * ties (cmpeq(min, m1)), renormalisation (subtract the minimum) only when metric[0] > 210.
 * ------------------------------------------------------------------------------------------ */
static inline int parity32(uint32_t v) { return __builtin_parity(v); }
static inline uint8_t adds_u8(unsigned a, unsigned b) { unsigned s = a + b; return (uint8_t)(s > 255 ? 255 : s); }

size_t orc_cc_decode_k7(const uint8_t* soft, size_t n, uint8_t* bits)
{
    orc_trace_event("cc_decode_k7()");
    static const int polys[2] = {109, 79};
    uint8_t branchtab[64];
    for (int s = 0; s < 32; s++)
        for (int j = 0; j < 2; j++) branchtab[j * 32 + s] = parity32((uint32_t)(2 * s) & (uint32_t)polys[j]) ? 255 : 0;
    uint8_t X[64], Y[64];
    uint64_t dec[86];
    int start = 0;
    size_t nblk = 0;
    while (160 * nblk + 172 <= n) {
        const uint8_t* syms = soft + 160 * nblk;
        for (int i = 0; i < 64; i++) X[i] = 63;
        X[start & 63] = 0;
        for (int s = 0; s < 86; s++) {
            uint64_t d = 0;
            for (int i = 0; i < 32; i++) {
                unsigned a = (unsigned)(branchtab[i] ^ syms[2 * s]);
                unsigned b = (unsigned)(branchtab[32 + i] ^ syms[2 * s + 1]);
                unsigned metric = (((a + b + 1) >> 1) >> 2) & 63u;
                uint8_t m0 = adds_u8(X[i], metric);
                uint8_t m1 = adds_u8(X[i + 32], 63u - metric);
                uint8_t m2 = adds_u8(X[i], 63u - metric);
                uint8_t m3 = adds_u8(X[i + 32], metric);
                uint8_t s0 = m1 < m0 ? m1 : m0;
                uint8_t s1 = m3 < m2 ? m3 : m2;
                uint64_t d0 = (s0 == m1), d1 = (s1 == m3);
                Y[2 * i] = s0;
                Y[2 * i + 1] = s1;
                d |= (d0 | (d1 << 1)) << (2 * i);
            }
            if (Y[0] > 210) {
                uint8_t mn = Y[0];
                for (int i = 1; i < 64; i++) if (mn > Y[i]) mn = Y[i];
                for (int i = 0; i < 64; i++) Y[i] = (uint8_t)(Y[i] - mn);
            }
            memcpy(X, Y, 64);
            dec[s] = d;
        }
        /* find_endstate: first minimum */
        int end = 0; uint8_t mn = X[0];
        for (int i = 1; i < 64; i++) if (X[i] < mn) { mn = X[i]; end = i; }
        /* chainback: steps 85..6 give bits 79..0; the state after 6 back-steps seeds the next block */
        int st = end, next = 0;
        uint8_t* out = bits + 80 * nblk;
        for (int nb = 79; nb >= 0; nb--) {
            int k = (int)((dec[nb + 6] >> st) & 1);
            st = (st >> 1) | (k << 5);
            out[nb] = (uint8_t)k;
            if (nb == 74) next = st;
        }
        start = next;
        nblk++;
    }
    return 80 * nblk;
}

/* lfsr [gr-digital include/gnuradio/digital/lfsr.h] */
void orc_descramble(const uint8_t* in, size_t n, uint32_t mask, uint32_t seed, int len, uint8_t* out)
{
    orc_trace_event("descramble(%u,%u,%d)", mask, seed, len);
    uint32_t sr = seed;
    for (size_t i = 0; i < n; i++) {
        uint32_t b = in[i] & 1u;
        out[i] = (uint8_t)(parity32(sr & mask) ^ b);
        sr = (sr >> 1) | (b << len);
    }
}
void orc_scramble(const uint8_t* in, size_t n, uint32_t mask, uint32_t seed, int len, uint8_t* out)
{
    orc_trace_event("scramble(%u,%u,%d)", mask, seed, len);
    uint32_t sr = seed;
    for (size_t i = 0; i < n; i++) {
        uint8_t o = (uint8_t)(sr & 1u);
        uint32_t nb = (uint32_t)parity32(sr & mask) ^ (in[i] & 1u);
        sr = (sr >> 1) | (nb << len);
        out[i] = o;
    }
}
/* cc_encoder streaming [gr-fec/lib/cc_encoder_impl.cc] */
void orc_cc_encode_k7(const uint8_t* bits, size_t n, uint8_t* out)
{
    orc_trace_event("cc_encode_k7()");
    uint32_t st = 0;
    for (size_t i = 0; i < n; i++) {
        st = (st << 1) | (bits[i] & 1u);
        out[2 * i] = (uint8_t)parity32(st & 109u);
        out[2 * i + 1] = (uint8_t)parity32(st & 79u);
    }
}

/* ------------------------------------------------------------------------------------------
 * clock_recovery_mm_cc [gr-digital/lib/clock_recovery_mm_cc_impl.cc], as instantiated by gr_demod_bpsk.cpp:56-62
 * (omega = sps, gain_omega = 2.5e-5, mu = 0.5, gain_mu = 0.05, omega_relative_limit = 0.001).  8-tap MMSE
 * interpolator (same table / summation order as symbol_sync); slicer_0deg gives (re > 0, im > 0) as 0/1.
 * A symbol at cursor ii exists iff in[ii..ii+7] exist (the block's scheduler FUDGE is not modelled).
 * ------------------------------------------------------------------------------------------ */
size_t orc_clock_recovery_mm_cc(const cf32* in, size_t n, float omega, float gain_omega, float mu, float gain_mu,
                                float omega_relative_limit, cf32* out)
{
    orc_trace_event("clock_recovery_mm_cc(%.9g,%.9g,%.9g,%.9g,%.9g)", omega, gain_omega, mu, gain_mu, omega_relative_limit);
    const float* T = orc_mmse_table();
    const float omega_mid = omega, omega_lim = omega_relative_limit * omega;
    cf32 p2 = {0, 0}, p1 = {0, 0}, p0 = {0, 0}, c2 = {0, 0}, c1 = {0, 0}, c0 = {0, 0};
    size_t ii = 0, oo = 0;
    while (ii + 8 <= n) {
        int imu = (int)rintf(mu * 128.0f);
        const float* t = T + imu * 8;
        cf32 y = {0.0f, 0.0f};
        for (int k = 0; k < 8; k++) {
            y.re = fmaf(t[7 - k], in[ii + (size_t)k].re, y.re);
            y.im = fmaf(t[7 - k], in[ii + (size_t)k].im, y.im);
        }
        p2 = p1; p1 = p0; p0 = y;
        c2 = c1; c1 = c0; c0.re = y.re > 0.0f ? 1.0f : 0.0f; c0.im = y.im > 0.0f ? 1.0f : 0.0f;
        /* x = (c0 - c2) * conj(p1); y = (p0 - p2) * conj(c1); mm = Re(y - x) */
        const float ar = c0.re - c2.re, ai = c0.im - c2.im;
        const float xr = ar * p1.re + ai * p1.im;
        const float br = p0.re - p2.re, bi = p0.im - p2.im;
        const float yr = br * c1.re + bi * c1.im;
        float mm = branchless_clip(yr - xr, 1.0f);
        out[oo++] = y;
        omega = omega + gain_omega * mm;
        if (fabsf(omega - omega_mid) > omega_lim) g_clamp_hits++;
        omega = omega_mid + branchless_clip(omega - omega_mid, omega_lim);
        mu = mu + omega + gain_mu * mm;
        const float fl = floorf(mu);
        ii += (size_t)(int)fl;
        mu = mu - fl;
    }
    return oo;
}

/* rssi_tag_block::work (reference src/gr/rssi_tag_block.cpp:43-68): every 300 samples one value
 * 10*log10f(sqrt(sum(|x|^4) / 300) + 1e-20) + calibration; the sum is a serial float accumulation.  Returns the count. */
size_t orc_rssi_tag(const cf32* in, size_t n, float calibration, float* db)
{
    float sum = 0.0f; int nitems = 0; size_t k = 0;
    for (size_t i = 0; i < n; i++) {
        const float pwr = in[i].re * in[i].re + in[i].im * in[i].im;
        sum += pwr * pwr;
        nitems += 1;
        if (nitems >= 300) {
            const float level = sqrtf(sum / (float)nitems);
            if (db) db[k] = (float)(10.0f * log10f(level + 1.0e-20f)) + calibration;
            k++;
            sum = 0; nitems = 0;
        }
    }
    return k;
}

/* ------------------------------------------------------------------------------------------
 * gr_deframer_bb (reference src/gr/gr_deframer_bb.cpp:24-48, 83-185): shift-register search for the sync words
 * (type 2: 0xB5; types 1 and 3: 0x89ED 0xED89 0x98DE 0xED77 0x8CC8, 24-bit 0x4C8A2B), then the sync bits and the next
 * bit_buf_len (64 / 32 / 384) bits go to the mailbox and the search restarts with a cleared register.
 * st[0] = shift register, st[1] = sync_found, st[2] = bit_buf_index carry over between calls (zero them to start).
 * Returns the number of bits appended to out (capacity >= 2 n + 24).
 * ------------------------------------------------------------------------------------------ */
static int deframer_find(int type, uint32_t reg, int* nbits)
{
    uint32_t temp = type != 2 ? (reg & 0xFFFF) : (reg & 0xFF);
    *nbits = type == 2 ? 8 : 16;
    if (type == 2 && temp == 0xB5) return (int)temp;
    if (temp == 0x89ED || temp == 0xED89 || temp == 0x98DE || temp == 0xED77 || temp == 0x8CC8) return (int)temp;
    temp = reg & 0xFFFFFF;
    if (temp == 0x4C8A2B) { if (type != 2) *nbits = 24; return (int)temp; }
    return 0;
}
size_t orc_deframer(int type, const uint8_t* bits, size_t n, uint32_t st[3], uint8_t* out)
{
    const uint32_t bit_buf_len = type == 1 ? 64 : type == 2 ? 32 : 384;
    size_t no = 0;
    for (size_t i = 0; i < n; i++) {
        if (!st[1]) {
            st[0] = (st[0] << 1) | (bits[i] & 1u);
            int nb; const int ft = deframer_find(type, st[0], &nb);
            if (ft) {
                st[1] = 1;
                for (int k = 0; k < nb; k++) out[no++] = (uint8_t)((ft >> (nb - 1 - k)) & 1);
                st[2] = 0;
                continue;
            }
        }
        if (st[1]) {
            out[no++] = bits[i] & 1u;
            st[2]++;
            if (st[2] >= bit_buf_len) { st[1] = 0; st[0] = 0; st[2] = 0; }
        }
    }
    return no;
}

/* ------------------------------------------------------------------------------------------
 * gr_modem::synchronize / findSync / packBytes (reference src/gr_modem.cpp:1119-1282, 980-994; frame types
 * src/layer1framing.h:8-24; mode table :203-322).  Per bit: search the mode's sync words in a shift register, then collect
 * the frame's bits, pack them MSB first and hand the frame to processReceivedData.  Here a frame becomes one record
 * { u32 frame_type, u32 nbytes | _modem_sync << 16, nbytes payload bytes padded to a multiple of 4 } appended to out
 * (_modem_sync = the counter at the moment the frame completes: what the voice gate of the 1k modes tests, :1376-1390).
 * st[0] = shift register, st[1] = sync_found, st[2] = bit_buf_index, st[3] = current frame type, st[4] = _modem_sync;
 * bitbuf (>= orc_modem_sync_geometry(...).bit_buf_len bytes) carries a partial frame between calls.  Returns bytes written.
 * cls 3 = M17 (gr_modem.cpp:1187-1210: 16-bit LSF / stream words first, else the 32-bit EOT word; 46-byte frames, :309-313).
 * ------------------------------------------------------------------------------------------ */
int orc_modem_sync_geometry(int modem_type, int* bit_buf_len, int* frame_length)
{
    int bits = 64, len = 7, cls = 2;   /* cls 0: 1k modes (0xB5), 1: fast modes (24-bit words only), 2: the rest, 3: M17 */
    switch (modem_type) {
    case 24: case 16: case 18: case 21: case 6: bits = 32; len = 4; cls = 0; break;              /* BPSK1K 2FSK1KFM 2FSK1K GMSK1K 4FSK1KFM */
    case 1: case 4: case 19: case 22: bits = 48 * 8; len = 47; break;                             /* QPSK20K 4FSK10KFM 2FSK10KFM GMSK10K */
    case 2: bits = 3123 * 8; len = 3122; cls = 1; break;                                          /* QPSKVideo */
    case 26: bits = 1517 * 8; len = 1516; cls = 1; break;                                         /* QPSK250K */
    case 27: bits = 623 * 8; len = 622; cls = 1; break;                                           /* 4FSK100K */
    case 40: bits = 46 * 8; len = 46; cls = 3; break;                                             /* M17, gr_modem.cpp:309-313 */
    default: break;                                                                               /* 2k modes: 64 bits, 7 bytes */
    }
    if (bit_buf_len) *bit_buf_len = bits;
    if (frame_length) *frame_length = len;
    return cls;
}
static uint32_t modem_find_sync(int cls, uint32_t reg)
{
    if (cls == 0) return (reg & 0xFF) == 0xB5 ? 0xB5u : 0u;                                       /* FrameTypeVoice1 */
    if (cls == 3) {                                                                               /* gr_modem.cpp:1187-1210 */
        if ((reg & 0xFFFF) == 0x55F7) return 0x55F7u;                                             /* FrameTypeM17LSF */
        if ((reg & 0xFFFF) == 0xFF5D) return 0xFF5Du;                                             /* FrameTypeM17Stream */
        return reg == 0x555D555Du ? 0x555D555Du : 0u;                                             /* FrameTypeM17EOT */
    }
    uint32_t t24 = reg & 0xFFFFFF;
    if (cls == 2) {
        if ((reg & 0xFFFF) == 0xED89) return 0xED89u;                                             /* FrameTypeVoice2 -> FrameTypeVoice */
        if (t24 == 0x89EDAA || t24 == 0xED77AA || t24 == 0x98DEAA || t24 == 0x8CC8DD || t24 == 0x4C8A2B) return t24;
        return 0;
    }
    if (t24 == 0xDE98AA || t24 == 0x98DEAA || t24 == 0x4C8A2B) return t24;                        /* IP, Video, End */
    return 0;
}
/* bits the last orc_modem_sync call collected into a frame while a sync was held: > 0 <=> gr_modem::synchronize returned data_to_process = true
 * (src/gr_modem.cpp:1121-1175); pinned against the reference's own gr_modem::demodulate() return value in tests/test_ref_modem.py */
static size_t g_modem_sync_collected = 0;
size_t orc_modem_sync_collected(void) { return g_modem_sync_collected; }
size_t orc_modem_sync(int modem_type, const uint8_t* bits, size_t n, uint32_t st[5], uint8_t* bitbuf, uint8_t* out)
{
    int bit_buf_len0, frame_length0;
    g_modem_sync_collected = 0;
    const int cls = orc_modem_sync_geometry(modem_type, &bit_buf_len0, &frame_length0);
    size_t no = 0;
    for (size_t i = 0; i < n; i++) {
        if (!st[1]) {
            st[0] = (st[0] << 1) | (bits[i] & 1u);
            const uint32_t ft = modem_find_sync(cls, st[0]);
            if (ft) {
                st[1] = 1; st[3] = ft; st[2] = 0;
                if (st[4] < 32) st[4] += 8;
                continue;
            }
            if (st[4] > 0) st[4] -= 1;
        }
        if (st[1]) {
            bitbuf[st[2]++] = bits[i] & 1u;
            g_modem_sync_collected++;
            int frame_length = frame_length0, bit_buf_len = bit_buf_len0;
            if ((cls == 1 || cls == 2) && st[3] == 0xED89) frame_length++;          /* reserved byte of voice frames */
            else if (cls == 1 || cls == 2) bit_buf_len = bit_buf_len0 - 8;
            if ((int)st[2] >= bit_buf_len) {
                uint32_t hdr[2] = {st[3], (uint32_t)frame_length | (st[4] << 16)};   /* nbytes | _modem_sync << 16 */
                memcpy(out + no, hdr, 8); no += 8;
                const size_t padded = ((size_t)frame_length + 3) & ~(size_t)3;
                memset(out + no, 0, padded);
                for (int k = 0; k < bit_buf_len; k += 8) {
                    int t = 0;
                    for (int j = 0; j < 8; j++) t = (t << 1) | (bitbuf[k + j] & 1);
                    out[no + (size_t)(k >> 3)] = (uint8_t)t;
                }
                no += padded;
                st[1] = 0; st[0] = 0; st[2] = 0;
            }
        }
    }
    return no;
}
