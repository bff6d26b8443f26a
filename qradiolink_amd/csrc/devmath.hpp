// devmath.hpp — device-side deterministic math shared by the HIP kernels.
// Compiled with -ffp-contract=off: every fused multiply-add is an explicit fmaf().
// The polynomial NCO replaces libm sincosf in rotator_cc / fll_band_edge_cc / costas_loop_cc
// (GNU Radio gr_expj) so that results do not depend on a math library; see DESIGN.md "Arithmetic".
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/qrl_contracts.h"

namespace qrl {

__device__ __forceinline__ float poly_sin(float x)
{
    const float z = x * x;
    float p = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    p = fmaf(z, p, -1.6666654611e-1f);
    return fmaf(x * z, p, x);
}
__device__ __forceinline__ float poly_cos(float x)
{
    const float z = x * x;
    float p = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    p = fmaf(z, p, 4.166664568298827e-2f);
    const float q = fmaf(z, -0.5f, 1.0f);
    return fmaf(z * z, p, q);
}
// returns (cos, sin)
// quadrant q: 0 -> (pc, ps), 1 -> (-ps, pc), 2 -> (-pc, -ps), 3 -> (ps, -pc).  Branch-free (two selects, two sign flips):
// the recursive loops call this once per sample and a switch costs four exec-mask regions on a wave that cannot hide them.
__device__ __forceinline__ float2 quad_fix(int q, float ps, float pc)
{
    const bool swap = q & 1;
    const uint32_t c0 = __float_as_uint(swap ? ps : pc), s0 = __float_as_uint(swap ? pc : ps);
    const uint32_t negc = ((uint32_t)(q + 1) & 2u) << 30, negs = ((uint32_t)q & 2u) << 30;
    return make_float2(__uint_as_float(c0 ^ negc), __uint_as_float(s0 ^ negs));
}
// (cos x, sin x) of a float angle in radians, |x| <~ 10
__device__ __forceinline__ float2 sincos_rad(float x)
{
    const float k = rintf(x * 0.636619772367581343f);
    float r = fmaf(-k, 1.57079637050628662109375f, x);
    r = fmaf(-k, -4.37113900018624283e-8f, r);
    return quad_fix((int)k, poly_sin(r), poly_cos(r));
}
// (cos, sin) of a 2^-64-turn fixed-point angle (top 32 bits used)
__device__ __forceinline__ float2 sincos_turn(uint64_t angle)
{
    const uint32_t a = (uint32_t)(angle >> 32);
    const uint32_t q = (a + 0x20000000u) >> 30;
    const int32_t r = (int32_t)(a - (q << 30));
    const float x = (float)r * 1.4629180792671596e-9f;
    return quad_fix((int)q, poly_sin(x), poly_cos(x));
}
// complex product with the fixed fmaf pattern of the rotator contract
__device__ __forceinline__ float2 cmul_fma(float2 a, float2 b)
{
    return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x));
}
// plain complex product (std::complex operator*: two products and one add/sub per part)
__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// gnuradio fast_atan2f with its 257-entry table T
__device__ __forceinline__ float fast_atan2f_lut(float y, float x, const float* __restrict__ T)
{
    const float y_abs = fabsf(y), x_abs = fabsf(x);
    if (!((y_abs > 0.0f) || (x_abs > 0.0f))) return 0.0f;
    const float z = (y_abs < x_abs) ? (y_abs / x_abs) : (x_abs / y_abs);
    float base;
    // gnuradio compares in double: (double)z < 0.003921569.  For a float z that is z < nextfloat-at-or-above(0.003921569) = 0x3B808082 exactly
    // (0.003921569 is not a float; the floats below it are < it, 0x3B808082 is the first one >= it): one f32 compare instead of a conversion and an f64 compare
    if (z < __uint_as_float(0x3B808082u)) {
        base = z;
    } else {
        float alpha = z * 255.0f;
        const int index = ((int)alpha) & 0xff;
        alpha -= (float)index;
        const float d = T[index + 1] - T[index];
        base = T[index] + d * alpha;
    }
    const float PI_F = 3.14159265358979323846f, PIO2_F = 1.57079632679489661923f;
    float angle;
    if (x_abs > y_abs) {
        if (x >= 0.0f) angle = (y >= 0.0f) ? base : -base;
        else           angle = (y >= 0.0f) ? (PI_F - base) : (base - PI_F);
    } else {
        if (y >= 0.0f) angle = (x >= 0.0f) ? (PIO2_F - base) : (PIO2_F + base);
        else           angle = (x >= 0.0f) ? (-PIO2_F + base) : (-PIO2_F - base);
    }
    return angle;
}

__device__ __forceinline__ float tanhf_lut(float x, const float* __restrict__ T)
{
    if (x > 2.0f) return 1.0f;
    if (x <= -2.0f) return -1.0f;
    int index = (int)(128.0f + 64.0f * x);
    index = index > 255 ? 255 : (index < 0 ? 0 : index);
    return T[index];
}

__device__ __forceinline__ float branchless_clip(float x, float clip)
{
    float x1 = fabsf(x + clip);
    const float x2 = fabsf(x - clip);
    x1 -= x2;
    return 0.5f * x1;
}

__device__ __forceinline__ float phase_wrap(float phase)
{
    const double TWO_PI = 6.283185307179586476925286766559;
    // same loops as gr::blocks::control_loop::phase_wrap; the common case (no lane of the wave out of range) is one uniform branch
    if (__builtin_amdgcn_ballot_w64(phase > (float)TWO_PI || phase < (float)(-TWO_PI)) == 0) return phase;
    while (phase > (float)TWO_PI) phase = (float)((double)phase - TWO_PI);
    while (phase < (float)(-TWO_PI)) phase = (float)((double)phase + TWO_PI);
    return phase;
}

// log2 of a float, the same bits on the host (oracle/orc_blocks.c orc_det_log2f) and on the device: libm's log2f is not the same
// function on both sides, IEEE double arithmetic in a fixed order is.  x = 2^e m, m in [sqrt(1/2), sqrt(2)), t = (m - 1) / (m + 1),
// ln m = 2 atanh t as an odd series to t^13 (|t| <= 0.172: truncation 4e-13), rounded to float once.  Stands in for log2f in
// nlog10_ff / volk_32fc_s32f_power_spectrum_32f (VOLK's log2f_non_ieee: +-inf becomes +-127).
__host__ __device__ inline float det_log2f(float x)
{
    if (!(x > 0.0f)) return x == 0.0f ? -127.0f : __builtin_nanf("");   // log2f(0) = -inf -> -127; negative -> nan
    if (x > 3.4028234e38f) return 127.0f;                                // log2f(inf) = inf -> 127
    int adj = 0;
    if (x < 1.17549435e-38f) { x *= 18446744073709551616.0f; adj = -64; }   // subnormal: scale by 2^64 (exact)
    unsigned bits = __builtin_bit_cast(unsigned, x);
    int e = (int)(bits >> 23) - 127 + adj;
    float m = __builtin_bit_cast(float, (bits & 0x007fffffu) | 0x3f800000u);   // [1, 2)
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

}  // namespace qrl
