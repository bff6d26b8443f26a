// kernels_ff.hip — feed-forward blocks at the decimated rate (<= 2 % of the input bytes).
// One thread per output item, inputs read from the producer's ring (L2-resident), one fmaf chain
// per output with k ascending: exactly the order of oracle/orc_blocks.c.
//   k_fir_ccf    fft_filter_ccf  (gr_demod_2fsk.cpp:91-92, gr_demod_gmsk.cpp:84-85, gr_demod_qpsk.cpp:100-103)
//   k_fir_fff    fft_filter_fff  (gr_demod_2fsk.cpp:104, gr_demod_gmsk.cpp:96-98)
//   k_quad_demod quadrature_demod_cf (gr_demod_gmsk.cpp:95, gr_demod_2fsk.cpp:112)
//   k_disc_2fsk  2x fft_filter_ccc + complex_to_mag + divide + rail(0,2) + add(-1) (gr_demod_2fsk.cpp:94-102,140-149)
#include "devmath.hpp"
#include "engine.hpp"

namespace qrl {

__device__ __forceinline__ float2 ringc_at(const RingC& r, int b, int64_t i)
{
    if (i < 0) return make_float2(0.f, 0.f);
    return r.p[(size_t)b * (r.mask + 1u) + ((uint32_t)i & r.mask)];
}
__device__ __forceinline__ float ringf_at(const RingF& r, int b, int64_t i)
{
    if (i < 0) return 0.f;
    return r.p[(size_t)b * (r.mask + 1u) + ((uint32_t)i & r.mask)];
}

__global__ __launch_bounds__(256) void k_fir_ccf(const FirCcfParams P)
{
    const int b = blockIdx.y;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= P.count) return;
    if (t == 0 && P.counts) P.counts[b * 4 + 0] = P.count;
    const int64_t n = (int64_t)(P.q0 + t);
    float ar = 0.f, ai = 0.f;
    for (int k = 0; k < P.nt; ++k) {
        const float h = P.taps[k];
        const float2 x = ringc_at(P.in, b, n - k);
        ar = fmaf(h, x.x, ar);
        ai = fmaf(h, x.y, ai);
    }
    const float2 y = make_float2(ar, ai);
    P.out.p[(size_t)b * (P.out.mask + 1u) + ((uint32_t)n & P.out.mask)] = y;
    if (P.port && t < P.port_cap) P.port[(size_t)b * P.port_cap + t] = y;
}
// LDS-tiled variants for the longer filters (RRC shaping filters of the FM / 4FSK / BPSK chains, 75 ... 501 taps): a workgroup
// stages FT_OUT + nt - 1 input items of one stream and the taps in LDS, every thread then produces FT_R outputs 256 apart
// (consecutive lanes -> consecutive LDS words, taps are wave-uniform broadcasts).  Same fmaf chain, k ascending.
constexpr int FT_R = 4, FT_OUT = 256 * FT_R, FT_MAXT = 1024;

__global__ __launch_bounds__(256) void k_fir_ccf_tiled(const FirCcfParams P)
{
    __shared__ float taps[FT_MAXT];
    __shared__ float2 xs[FT_OUT + FT_MAXT];
    const int b = blockIdx.y, tid = threadIdx.x;
    const uint32_t t0 = blockIdx.x * (uint32_t)FT_OUT;
    const int nt = P.nt;
    for (int k = tid; k < nt; k += 256) taps[k] = P.taps[k];
    const int64_t first = (int64_t)(P.q0 + t0) - (nt - 1);         // xs[i] = x[first + i]
    const int span = min((uint32_t)FT_OUT, P.count - t0) + nt - 1;
    for (int i = tid; i < span; i += 256) xs[i] = ringc_at(P.in, b, first + i);
    __syncthreads();
    if (blockIdx.x == 0 && tid == 0 && P.counts) P.counts[b * 4 + 0] = P.count;
    float ar[FT_R], ai[FT_R];
#pragma unroll
    for (int r = 0; r < FT_R; ++r) { ar[r] = 0.f; ai[r] = 0.f; }
    const float2* xp = xs + (nt - 1) + tid;                        // output j = tid + 256 r reads xp[256 r - k]
    for (int k = 0; k < nt; ++k) {
        const float h = taps[k];
#pragma unroll
        for (int r = 0; r < FT_R; ++r) {
            const float2 x = xp[256 * r - k];
            ar[r] = fmaf(h, x.x, ar[r]);
            ai[r] = fmaf(h, x.y, ai[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < FT_R; ++r) {
        const uint32_t t = t0 + tid + 256u * r;
        if (t < P.count) {
            const int64_t n = (int64_t)(P.q0 + t);
            const float2 y = make_float2(ar[r], ai[r]);
            P.out.p[(size_t)b * (P.out.mask + 1u) + ((uint32_t)n & P.out.mask)] = y;
            if (P.port && t < P.port_cap) P.port[(size_t)b * P.port_cap + t] = y;
        }
    }
}
__global__ __launch_bounds__(256) void k_fir_fff_tiled(const FirFffParams P)
{
    __shared__ float taps[FT_MAXT];
    __shared__ float xs[FT_OUT + FT_MAXT];
    const int b = blockIdx.y, tid = threadIdx.x;
    const uint32_t t0 = blockIdx.x * (uint32_t)FT_OUT;
    const int nt = P.nt;
    for (int k = tid; k < nt; k += 256) taps[k] = P.taps[k];
    const int64_t first = (int64_t)(P.q0 + t0) - (nt - 1);
    const int span = min((uint32_t)FT_OUT, P.count - t0) + nt - 1;
    for (int i = tid; i < span; i += 256) xs[i] = ringf_at(P.in, b, first + i);
    __syncthreads();
    float a[FT_R];
#pragma unroll
    for (int r = 0; r < FT_R; ++r) a[r] = 0.f;
    const float* xp = xs + (nt - 1) + tid;
    for (int k = 0; k < nt; ++k) {
        const float h = taps[k];
#pragma unroll
        for (int r = 0; r < FT_R; ++r) a[r] = fmaf(h, xp[256 * r - k], a[r]);
    }
#pragma unroll
    for (int r = 0; r < FT_R; ++r) {
        const uint32_t t = t0 + tid + 256u * r;
        if (t < P.count) P.out.p[(size_t)b * (P.out.mask + 1u) + ((uint32_t)(P.q0 + t) & P.out.mask)] = a[r];
    }
}
static bool fir_use_tiled(int nt, uint32_t count) { return nt >= 32 && nt <= FT_MAXT && count >= 512; }

void launch_fir_ccf(const FirCcfParams& p, int batch, hipStream_t s)
{
    if (!p.count) return;
    if (fir_use_tiled(p.nt, p.count)) hipLaunchKernelGGL(k_fir_ccf_tiled, dim3((p.count + FT_OUT - 1) / FT_OUT, batch), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(k_fir_ccf, dim3((p.count + 255) / 256, batch), dim3(256), 0, s, p);
}

__global__ __launch_bounds__(256) void k_fir_fff(const FirFffParams P)
{
    const int b = blockIdx.y;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= P.count) return;
    const int64_t n = (int64_t)(P.q0 + t);
    float a = 0.f;
    for (int k = 0; k < P.nt; ++k) a = fmaf(P.taps[k], ringf_at(P.in, b, n - k), a);
    P.out.p[(size_t)b * (P.out.mask + 1u) + ((uint32_t)n & P.out.mask)] = a;
}
void launch_fir_fff(const FirFffParams& p, int batch, hipStream_t s)
{
    if (!p.count) return;
    if (fir_use_tiled(p.nt, p.count)) hipLaunchKernelGGL(k_fir_fff_tiled, dim3((p.count + FT_OUT - 1) / FT_OUT, batch), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(k_fir_fff, dim3((p.count + 255) / 256, batch), dim3(256), 0, s, p);
}

__global__ __launch_bounds__(256) void k_quad_demod(const QuadDemodParams P)
{
    __shared__ float T[257];
    for (int k = threadIdx.x; k < 257; k += 256) T[k] = P.atan_tab[k];
    __syncthreads();
    const int b = blockIdx.y;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= P.count) return;
    const int64_t n = (int64_t)(P.q0 + t);
    const float2 a = ringc_at(P.in, b, n), p = ringc_at(P.in, b, n - 1);
    const float re = a.x * p.x + a.y * p.y;
    const float im = a.y * p.x - a.x * p.y;
    P.out.p[(size_t)b * (P.out.mask + 1u) + ((uint32_t)n & P.out.mask)] = P.gain * fast_atan2f_lut(im, re, T);
}
void launch_quad_demod(const QuadDemodParams& p, int batch, hipStream_t s)
{
    if (!p.count) return;
    hipLaunchKernelGGL(k_quad_demod, dim3((p.count + 255) / 256, batch), dim3(256), 0, s, p);
}

__global__ __launch_bounds__(256) void k_disc_2fsk(const Disc2fskParams P)
{
    const int b = blockIdx.y;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= P.count) return;
    const int64_t n = (int64_t)(P.q0 + t);
    float ur = 0.f, ui = 0.f, lr = 0.f, li = 0.f;
    for (int k = 0; k < P.nt; ++k) {
        const float2 x = ringc_at(P.in, b, n - k);
        const float2 hu = P.up[k], hl = P.lo[k];
        ur = fmaf(hu.x, x.x, ur); ur = fmaf(-hu.y, x.y, ur);
        ui = fmaf(hu.x, x.y, ui); ui = fmaf(hu.y, x.x, ui);
        lr = fmaf(hl.x, x.x, lr); lr = fmaf(-hl.y, x.y, lr);
        li = fmaf(hl.x, x.y, li); li = fmaf(hl.y, x.x, li);
    }
    const float mu = sqrtf(ur * ur + ui * ui);
    const float ml = sqrtf(lr * lr + li * li);
    float r = mu / ml;
    if (!(r >= 0.0f)) r = 0.0f;   // rail_ff lower bound; NaN (0/0) -> 0
    if (r > 2.0f) r = 2.0f;
    P.out.p[(size_t)b * (P.out.mask + 1u) + ((uint32_t)n & P.out.mask)] = r + (-1.0f);
}
// gr_demod_4fsk non-FM branch (gr_demod_4fsk.cpp:110-127,165-176 + gr_4fsk_discriminator::work, gr_4fsk_discriminator.cpp:17-44):
// four complex band-pass filters -> complex_to_mag -> strict arg-max -> one of (+-0.707107, +-0.707107), else 0
__global__ __launch_bounds__(256) void k_disc_4fsk(const Disc4fskParams P)
{
    const int b = blockIdx.y;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= P.count) return;
    const int64_t n = (int64_t)(P.q0 + t);
    float re[4] = {0.f, 0.f, 0.f, 0.f}, im[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < P.nt; ++k) {
        const float2 x = ringc_at(P.in, b, n - k);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float2 h = P.taps[q * P.nt + k];
            re[q] = fmaf(h.x, x.x, re[q]); re[q] = fmaf(-h.y, x.y, re[q]);
            im[q] = fmaf(h.x, x.y, im[q]); im[q] = fmaf(h.y, x.x, im[q]);
        }
    }
    const float m1 = sqrtf(re[0] * re[0] + im[0] * im[0]), m2 = sqrtf(re[1] * re[1] + im[1] * im[1]);
    const float m3 = sqrtf(re[2] * re[2] + im[2] * im[2]), m4 = sqrtf(re[3] * re[3] + im[3] * im[3]);
    const float A = 0.707107f;
    float2 v = make_float2(0.f, 0.f);
    if (m1 > m2 && m1 > m3 && m1 > m4) v = make_float2(-A, -A);
    else if (m2 > m1 && m2 > m3 && m2 > m4) v = make_float2(-A, A);
    else if (m3 > m2 && m3 > m1 && m3 > m4) v = make_float2(A, A);
    else if (m4 > m2 && m4 > m1 && m4 > m3) v = make_float2(A, -A);
    P.out.p[(size_t)b * (P.out.mask + 1u) + ((uint32_t)n & P.out.mask)] = v;
}
void launch_disc_4fsk(const Disc4fskParams& p, int batch, hipStream_t s)
{
    if (!p.count) return;
    hipLaunchKernelGGL(k_disc_4fsk, dim3((p.count + 255) / 256, batch), dim3(256), 0, s, p);
}

void launch_disc_2fsk(const Disc2fskParams& p, int batch, hipStream_t s)
{
    if (!p.count) return;
    hipLaunchKernelGGL(k_disc_2fsk, dim3((p.count + 255) / 256, batch), dim3(256), 0, s, p);
}

// ---- fused feed-forward part of the non-FM 2FSK chain (gr_demod_2fsk.cpp:91-104,140-152):
//   _filter (fft_filter_ccf) -> {_upper_filter, _lower_filter} (fft_filter_ccc) -> complex_to_mag x2 -> divide ->
//   rail(0,2) -> add_const(-1) -> _symbol_filter (fft_filter_fff)
// One workgroup = 256 consecutive outputs of one stream; the three FIR stages run out of LDS with the
// intermediate halos recomputed (12 % extra MACs), taps are wave-uniform (scalar loads), every output is the
// same fmaf chain (k ascending) as the unfused kernels.  Items at negative absolute index are zero, exactly
// as a ring read in front of the stream start returns zero.
constexpr int FF_T = 256;
__global__ __launch_bounds__(256) void k_2fsk_ff(const Fsk2FfParams P)
{
    __shared__ float2 lt[FF_T + 104 + 8];   // FLL output, abs = n0t - (nf-1) - (nb-1) - (ns-1) + i
    __shared__ float2 ft[FF_T + 64 + 8];    // _filter output
    __shared__ float dt[FF_T + 24 + 8];     // discriminator output
    const int b = blockIdx.y, tid = threadIdx.x;
    const int nf = P.nf, nb = P.nb, ns = P.ns;           // 41, 41, 25 for the 1k mode
    const int hf = nf - 1, hb = nb - 1, hs = ns - 1;
    const int64_t n0t = (int64_t)P.q0 + (int64_t)blockIdx.x * FF_T;
    const int nl = FF_T + hf + hb + hs, nfo = FF_T + hb + hs, nd = FF_T + hs;
    for (int i = tid; i < nl; i += 256) lt[i] = ringc_at(P.in, b, n0t - (hf + hb + hs) + i);
    __syncthreads();
    for (int j = tid; j < nfo; j += 256) {                 // f at abs = n0t - (hb + hs) + j
        float ar = 0.f, ai = 0.f;
        for (int k = 0; k < nf; ++k) {
            const float h = P.tf[k];
            const float2 x = lt[j + hf - k];
            ar = fmaf(h, x.x, ar);
            ai = fmaf(h, x.y, ai);
        }
        const bool neg = n0t - (hb + hs) + j < 0;
        ft[j] = neg ? make_float2(0.f, 0.f) : make_float2(ar, ai);
    }
    __syncthreads();
    for (int j = tid; j < nd; j += 256) {                  // d at abs = n0t - hs + j
        float ur = 0.f, ui = 0.f, lr = 0.f, li = 0.f;
        for (int k = 0; k < nb; ++k) {
            const float2 x = ft[j + hb - k];
            const float2 hu = P.up[k], hl = P.lo[k];
            ur = fmaf(hu.x, x.x, ur); ur = fmaf(-hu.y, x.y, ur);
            ui = fmaf(hu.x, x.y, ui); ui = fmaf(hu.y, x.x, ui);
            lr = fmaf(hl.x, x.x, lr); lr = fmaf(-hl.y, x.y, lr);
            li = fmaf(hl.x, x.y, li); li = fmaf(hl.y, x.x, li);
        }
        const float mu = sqrtf(ur * ur + ui * ui);
        const float ml = sqrtf(lr * lr + li * li);
        float r = mu / ml;
        if (!(r >= 0.0f)) r = 0.0f;   // rail_ff lower bound; NaN (0/0) -> 0
        if (r > 2.0f) r = 2.0f;
        dt[j] = (n0t - hs + j < 0) ? 0.f : r + (-1.0f);
    }
    __syncthreads();
    const uint32_t t = blockIdx.x * (uint32_t)FF_T + tid;   // output index inside this call
    if (t < P.count) {
        float a = 0.f;
        for (int k = 0; k < ns; ++k) a = fmaf(P.ts[k], dt[tid + hs - k], a);
        const int64_t n = n0t + tid;
        P.out.p[(size_t)b * (P.out.mask + 1u) + ((uint32_t)n & P.out.mask)] = a;
        if (t == 0 && P.counts) P.counts[b * 4 + 0] = P.count;
        if (P.port && t < P.port_cap) P.port[(size_t)b * P.port_cap + t] = ft[tid + hb + hs];
    }
}
void launch_2fsk_ff(const Fsk2FfParams& p, int batch, hipStream_t s)
{
    if (!p.count) return;
    hipLaunchKernelGGL(k_2fsk_ff, dim3((p.count + FF_T - 1) / FF_T, batch), dim3(256), 0, s, p);
}

}  // namespace qrl
