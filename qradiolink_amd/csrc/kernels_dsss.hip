// kernels_dsss.hip — the DSSS mode "BPSK 8" behind its two resamplers (reference src/gr/gr_demod_dsss.cpp:30-111, instance
// make_gr_demod_dsss(25, 1000000, 1700, 150) gr_demod_base.cpp:218), everything at 5 200 samples/s and below:
//   k_dsss_loop<0>  costas_loop_cc(pi / 200, 2, use_snr = true)   (_costas_freq, in front of the channel filter)
//   k_dsss_loop<1>  agc2_cc(0.1, 0.1, 1, 10)                      (_agc, behind the channel filter)
//   k_dsss_mf       dsss_decoder_cc: Barker-13 matched filter (600 taps), the largest of the 325 evaluations of a code period
//                   (src/gr/dsss_decoder_cc_impl.cc:128-167)
//   k_dsss_tail     clock_recovery_mm_cc(1, 2.5e-5, 0.5, 0.05, 0.005) -> costas_loop_cc(2 pi / 100, 2) -> real part x 64 + 128
// 16 symbols per second and stream: one lane per stream straight out of the rings, no staging -- the rates are five to seven
// orders of magnitude below the front end's.  Arithmetic = oracle/orc_chains.c orc_demod_dsss, bit for bit.
#include "devmath.hpp"
#include "engine.hpp"

namespace qrl {

__device__ __forceinline__ float2 ring_at(const RingC& r, int b, int64_t i)
{
    if (i < 0) return make_float2(0.f, 0.f);
    return r.p[(size_t)b * (r.mask + 1u) + ((uint32_t)i & r.mask)];
}

template <int MODE>
__global__ __launch_bounds__(64) void k_dsss_loop(const DsssLoopParams P, int batch)
{
    __shared__ float th[256];
    for (int k = threadIdx.x; k < 256; k += 64) th[k] = P.tanh_tab[k];
    __syncthreads();
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= batch) return;
    DsssState st = P.st[b];
    float2* out = P.out.p + (size_t)b * (P.out.mask + 1u);
    for (uint32_t t = 0; t < P.count; ++t) {
        const uint64_t n = P.q0 + t;
        const float2 x = ring_at(P.in, b, (int64_t)n);
        float2 o;
        if (MODE == 0) {   // costas_loop_cc, order 2, SNR-weighted error (contract: oracle orc_costas)
            const float2 nco = sincos_rad(-st.phase);
            o.x = x.x * nco.x - x.y * nco.y; o.y = x.x * nco.y + x.y * nco.x;
            const float snr = (o.x * o.x + o.y * o.y);
            float e = tanhf_lut(snr * o.x, th) * o.y;
            e = branchless_clip(e, 1.0f);
            st.freq = st.freq + P.beta * e;
            st.phase = st.phase + st.freq + P.alpha * e;
            st.phase = phase_wrap(st.phase);
            if (st.freq > 1.0f) st.freq = 1.0f; else if (st.freq < -1.0f) st.freq = -1.0f;
        } else {           // agc2_cc(0.1, 0.1, 1, gain), max gain 65536 (contract: oracle orc_agc2)
            o.x = x.x * st.gain; o.y = x.y * st.gain;
            const float tmp = -1.0f + sqrtf(o.x * o.x + o.y * o.y);
            st.gain -= tmp * 0.1f;
            if (st.gain < 0.0f) st.gain = 10e-5f;
            if (st.gain > 65536.0f) st.gain = 65536.0f;
        }
        out[(uint32_t)n & P.out.mask] = o;
    }
    P.st[b] = st;
}
void launch_dsss_loop(const DsssLoopParams& p, int mode, int batch, hipStream_t s)
{
    if (!p.count) return;
    if (mode == 0) hipLaunchKernelGGL(k_dsss_loop<0>, dim3((batch + 63) / 64), dim3(64), 0, s, p, batch);
    else hipLaunchKernelGGL(k_dsss_loop<1>, dim3((batch + 63) / 64), dim3(64), 0, s, p, batch);
}

// one wave per (output, stream): the 925 samples the 325 windows of a code period cover are staged in LDS once
constexpr int DS_L = 325, DS_NT = 600;
__global__ __launch_bounds__(64) void k_dsss_mf(const DsssMfParams P)
{
    __shared__ float taps[DS_NT];
    __shared__ float2 xs[DS_L + DS_NT];
    const int lane = threadIdx.x, b = blockIdx.y;
    const uint64_t I = P.i0 + blockIdx.x;
    for (int k = lane; k < DS_NT; k += 64) taps[k] = P.taps[k];
    const int64_t P0 = (int64_t)DS_L * ((int64_t)I - 2) + 1;      // window j covers x[P0 + j .. P0 + j + 600): history 325 = 324 old items + the new one
    for (int i = lane; i < DS_L + DS_NT - 1; i += 64) xs[i] = ring_at(P.in, b, P0 + i);
    __syncthreads();
    float best = 0.0f; float2 bv = make_float2(0.f, 0.f); int bj = 0x7fffffff;
    for (int j = lane; j < DS_L; j += 64) {
        float ar = 0.f, ai = 0.f;
        for (int k = 0; k < DS_NT; ++k) {
            const float2 x = xs[j + DS_NT - 1 - k];
            ar = fmaf(taps[k], x.x, ar);
            ai = fmaf(taps[k], x.y, ai);
        }
        const float a2 = ar * ar, b2 = ai * ai;
        const float cur = sqrtf(a2 + b2);
        if (cur > best) { best = cur; bv = make_float2(ar, ai); bj = j; }   // the first maximum of this lane's (ascending) evaluations
    }
    // the first maximum over all evaluations: largest magnitude, smallest j among equals (a lane that never beat 0 carries j = max)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off, 64); const int oj = __shfl_xor(bj, off, 64);
        const float ox = __shfl_xor(bv.x, off, 64), oy = __shfl_xor(bv.y, off, 64);
        if (ob > best || (ob == best && oj < bj)) { best = ob; bj = oj; bv = make_float2(ox, oy); }
    }
    if (lane == 0) {
        const float sc = 2.0f / (float)DS_L;
        P.out.p[(size_t)b * (P.out.mask + 1u) + ((uint32_t)I & P.out.mask)] = make_float2(bv.x * sc, bv.y * sc);
    }
}
void launch_dsss_mf(const DsssMfParams& p, int batch, hipStream_t s)
{
    if (!p.count) return;
    hipLaunchKernelGGL(k_dsss_mf, dim3(p.count, batch), dim3(64), 0, s, p);
}

__global__ __launch_bounds__(64) void k_dsss_tail(const DsssTailParams P, int batch)
{
    __shared__ float mm[129 * 8];
    for (int k = threadIdx.x; k < 129 * 8; k += 64) mm[k] = P.mmse[k];
    __syncthreads();
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= batch) return;
    DsssTailState st = P.st[b];
    const uint64_t oo0 = st.oo;
    uint8_t* soft = P.soft.p + (size_t)b * (P.soft.mask + 1u);
    while (st.ii + 8 <= P.avail) {
        // clock_recovery_mm_cc (oracle orc_clock_recovery_mm_cc), omega starts at 1 sample per symbol
        const int imu = (int)rintf(st.mu * 128.0f);
        const float* t = mm + imu * 8;
        float2 y = make_float2(0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float2 x = ring_at(P.in, b, (int64_t)(st.ii + k));
            y.x = fmaf(t[7 - k], x.x, y.x);
            y.y = fmaf(t[7 - k], x.y, y.y);
        }
        st.p2 = st.p1; st.p1 = st.p0; st.p0 = y;
        st.c2 = st.c1; st.c1 = st.c0;
        st.c0.x = y.x > 0.f ? 1.0f : 0.0f; st.c0.y = y.y > 0.f ? 1.0f : 0.0f;
        const float ar = st.c0.x - st.c2.x, ai = st.c0.y - st.c2.y;
        const float xr = ar * st.p1.x + ai * st.p1.y;
        const float br = st.p0.x - st.p2.x, bi = st.p0.y - st.p2.y;
        const float yr = br * st.c1.x + bi * st.c1.y;
        const float mmv = branchless_clip(yr - xr, 1.0f);
        st.omega = st.omega + P.gain_omega * mmv;
        st.omega = P.omega_mid + branchless_clip(st.omega - P.omega_mid, P.omega_lim);
        st.mu = st.mu + st.omega + P.gain_mu * mmv;
        const float fl = floorf(st.mu);
        st.ii += (uint64_t)(int)fl;
        st.mu = st.mu - fl;
        // costas_loop_cc(2 pi / 100, 2), use_snr = false
        const float2 nco = sincos_rad(-st.phase);
        float2 o; o.x = y.x * nco.x - y.y * nco.y; o.y = y.x * nco.y + y.y * nco.x;
        float e = o.x * o.y;
        e = branchless_clip(e, 1.0f);
        st.freq = st.freq + P.beta * e;
        st.phase = st.phase + st.freq + P.alpha * e;
        st.phase = phase_wrap(st.phase);
        if (st.freq > 1.0f) st.freq = 1.0f; else if (st.freq < -1.0f) st.freq = -1.0f;
        // complex_to_real -> multiply_const(64) -> add_const(128) -> float_to_uchar
        float q = o.x * 64.0f; q = q + 128.0f;
        float r = rintf(q);
        if (!(r >= 0.f)) r = 0.f; if (r > 255.f) r = 255.f;
        soft[(uint32_t)st.oo & P.soft.mask] = (uint8_t)r;
        const uint64_t kk = st.oo - oo0;
        if (P.port && kk < P.port_cap) P.port[(size_t)b * P.port_cap + kk] = o;
        st.oo++;
    }
    P.st[b] = st;
    P.counts[b * 4 + 1] = (uint32_t)(st.oo - oo0);
}
void launch_dsss_tail(const DsssTailParams& p, int batch, hipStream_t s)
{
    hipLaunchKernelGGL(k_dsss_tail, dim3((batch + 63) / 64), dim3(64), 0, s, p, batch);
}

}  // namespace qrl
