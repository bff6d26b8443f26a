// kernels_ff.hip — feed-forward blocks at the decimated rate (<= 2 % of the input bytes).
// One thread per output item, inputs read from the producer's ring (L2-resident), one fmaf chain
// per output with k ascending: exactly the order of oracle/orc_blocks.c.
//   k_fir_ccf    fft_filter_ccf  (gr_demod_2fsk.cpp:91-92, gr_demod_gmsk.cpp:84-85, gr_demod_qpsk.cpp:100-103)
//   k_fir_fff    fft_filter_fff  (gr_demod_2fsk.cpp:104, gr_demod_gmsk.cpp:96-98)
//   k_quad_demod quadrature_demod_cf (gr_demod_gmsk.cpp:95, gr_demod_2fsk.cpp:112)
//   k_disc_2fsk  2x fft_filter_ccc + complex_to_mag + divide + rail(0,2) + add(-1) (gr_demod_2fsk.cpp:94-102,140-149)
#include "devmath.hpp"
#include "engine.hpp"
#include <cstdlib>

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

template <int MAXT>   // 1024 for the digital chains; 2048 for the 1904-tap channel filter of gr_demod_wbfm (its own instance: the larger LDS
                      // footprint would cost the short filters occupancy)
__global__ __launch_bounds__(256) void k_fir_ccf_tiled(const FirCcfParams P)
{
    __shared__ float taps[MAXT];
    __shared__ float2 xs[FT_OUT + MAXT];
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
// (the per-thread global-load form only for the shortest filters / calls: at 23 taps it is texture-address bound -- C5: 2.57 ms
// for 134 M outputs, the tiled form reads every input once)
static bool fir_use_tiled(int nt, uint32_t count) { return nt >= 8 && nt <= FT_MAXT && count >= 512; }

void launch_fir_ccf(const FirCcfParams& p, int batch, hipStream_t s)
{
    if (!p.count) return;
    if (fir_use_tiled(p.nt, p.count)) hipLaunchKernelGGL(k_fir_ccf_tiled<FT_MAXT>, dim3((p.count + FT_OUT - 1) / FT_OUT, batch), dim3(256), 0, s, p);
    else if (p.nt > FT_MAXT && p.nt <= 2 * FT_MAXT && p.count >= 512)
        hipLaunchKernelGGL(k_fir_ccf_tiled<2 * FT_MAXT>, dim3((p.count + FT_OUT - 1) / FT_OUT, batch), dim3(256), 0, s, p);
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
    const float ang = fast_atan2f_lut(im, re, T);
    P.out.p[(size_t)b * (P.out.mask + 1u) + ((uint32_t)n & P.out.mask)] = P.gain * ang;
    if (P.out2.p) P.out2.p[(size_t)b * (P.out2.mask + 1u) + ((uint32_t)n & P.out2.mask)] = P.gain2 * ang;
    if (P.s16) {   // multiply_const_ff(level) + float_to_short(1, scale) (gr_demod_mmdvm_multi2.cpp:84,92), as k_f2s
        float r = rintf(((P.gain * ang) * P.s16_level) * P.s16_scale);
        if (r > 32767.0f) r = 32767.0f;
        if (r < -32768.0f) r = -32768.0f;
        if (t < P.s16_cap) P.s16[(size_t)b * P.s16_cap + t] = (int16_t)r;
        if (t == 0 && P.s16_counts) P.s16_counts[b] = P.count < P.s16_cap ? P.count : (uint32_t)P.s16_cap;
    }
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
    // the two filters are a conjugate pair (the engine checks it bit for bit): h = a + j b, A = sum a x, B = sum b x (real taps, k
    // ascending), upper = (A.re - B.im, A.im + B.re), lower = (A.re + B.im, A.im - B.re) -- oracle orc_fir_ccc_conj_pair
    float ar = 0.f, ai = 0.f, br = 0.f, bi = 0.f;
    for (int k = 0; k < P.nt; ++k) {
        const float2 x = ringc_at(P.in, b, n - k);
        const float2 h = P.up[k];
        ar = fmaf(h.x, x.x, ar); ai = fmaf(h.x, x.y, ai);
        br = fmaf(h.y, x.x, br); bi = fmaf(h.y, x.y, bi);
    }
    const float ur = ar - bi, ui = ai + br, lr = ar + bi, li = ai - br;
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
// One workgroup = FF_T consecutive outputs of one stream; the three FIR stages run out of LDS with the intermediate
// halos recomputed (6 % extra MACs).  Every thread owns FOUR consecutive outputs of a stage and slides a register
// window over the taps (one 8-byte LDS read per tap for four outputs instead of four); tiles are stored TRANSPOSED
// by 4 (item i at row i & 3, word (i >> 2) + 1) so that the lanes of a wave read consecutive LDS words.  Taps are
// wave-uniform (scalar loads) and zero padded to 4A + 1 entries: every output is the same fmaf chain (k ascending) as
// the unfused kernels followed by zero taps, which leave the value untouched.  Items at negative absolute index are
// zero, exactly as a ring read in front of the stream start returns zero.
constexpr int FF_PL = 280, FF_PF = 264, FF_PD = 264;   // row pitches (words); 8-byte rows: pitch = 8 or 24 (mod 32)

// acc[r] += sum_k taps[k] * tile[4 t + r + 4 A - k], k = 0 .. 4 nq - 1, fmaf chain k ascending (CPLX: complex taps)
// NQ > 0: the number of tap quads is a compile-time constant and the loop is unrolled by four -- the (wave-uniform, scalar) tap loads of four
// quads are issued together, one wait per sixteen taps; NQ = 0: run-time count, one scalar load and one wait per quad.  The scalar loads share
// their counter with the LDS reads, so they cannot be prefetched across a quad's LDS wait: at C1's geometry (41 + 41 + 25 taps) the per-quad
// form waits 29 times per thread, 1 011 us alone against 855 (a complete unroll: 914 us and 105 VGPRs).  profiles/r06_c1_helper_stream.log
template <int PITCH, int NQ, typename TapT, typename ItemT, typename Fma>
__device__ __forceinline__ void ff_window4(const ItemT* __restrict__ tile, int w, int nq_rt, const TapT* __restrict__ taps, Fma&& fma4)
{
    ItemT cur[4], nxt[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) cur[s] = tile[s * PITCH + w];
    auto quad = [&](int q) {
        --w;
#pragma unroll
        for (int s = 0; s < 4; ++s) nxt[s] = tile[s * PITCH + w];   // word w - 1: subs 1..3 are this step's u < 0 samples
        const TapT h0 = taps[4 * q], h1 = taps[4 * q + 1], h2 = taps[4 * q + 2], h3 = taps[4 * q + 3];
        fma4(h0, cur[0], cur[1], cur[2], cur[3]);     // e = 0: output r takes sample u = r
        fma4(h1, nxt[3], cur[0], cur[1], cur[2]);     // e = 1: u = r - 1
        fma4(h2, nxt[2], nxt[3], cur[0], cur[1]);     // e = 2: u = r - 2
        fma4(h3, nxt[1], nxt[2], nxt[3], cur[0]);     // e = 3: u = r - 3
#pragma unroll
        for (int s = 0; s < 4; ++s) cur[s] = nxt[s];
    };
    if constexpr (NQ > 0) {
#pragma unroll 4
        for (int q = 0; q < NQ; ++q) quad(q);
    } else {
        for (int q = 0; q < nq_rt; ++q) quad(q);
    }
}

template <int NQF, int NQB, int NQS>   // tap quads of the three filters ((padded taps - 1) / 4 + 1), or 0, 0, 0 = run-time counts
__global__ __launch_bounds__(256) void k_2fsk_ff(const Fsk2FfParams P)
{
    __shared__ float2 lt[4 * FF_PL];   // FLL output, item i <-> abs n0t - (hf + hb + hs) + i
    __shared__ float2 ft[4 * FF_PF];   // _filter output, item j <-> abs n0t - (hb + hs) + j
    __shared__ float dt[4 * FF_PD];    // discriminator output, item j <-> abs n0t - hs + j
    const int b = blockIdx.y, tid = threadIdx.x;
    const int hf = P.nf - 1, hb = P.nb - 1, hs = P.ns - 1;      // multiples of 4 (padded tap counts 4A + 1)
    const int T = 1024 - hb - hs;                                // outputs per workgroup: stage 1 computes exactly 1024 items
    const int64_t n0t = (int64_t)P.q0 + (int64_t)blockIdx.x * T;
    const int nl = 1024 + hf, nd = T + hs;
    if (tid < 4) { lt[tid * FF_PL] = make_float2(0.f, 0.f); ft[tid * FF_PF] = make_float2(0.f, 0.f); dt[tid * FF_PD] = 0.f; }
    {   // nl <= 1064 items: five unconditional ring reads per thread in flight together (ring reads are in bounds for any index)
        float2 v[5];
#pragma unroll
        for (int it = 0; it < 5; ++it) {
            const int64_t a = n0t - (hf + hb + hs) + tid + 256 * it;
            v[it] = P.in.p[(size_t)b * (P.in.mask + 1u) + ((uint32_t)a & P.in.mask)];
            if (a < 0) v[it] = make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int it = 0; it < 5; ++it) {
            const int i = tid + 256 * it;
            if (i < nl) lt[(i & 3) * FF_PL + (i >> 2) + 1] = v[it];
        }
    }
    __syncthreads();
    {   // stage 1: f[j] = sum tf[k] l[j + hf - k], j = 4 tid + r
        float2 a[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
        ff_window4<FF_PL, NQF>(lt, tid + (hf >> 2) + 1, (hf >> 2) + 1, P.tf,
                          [&](float h, const float2& x0, const float2& x1, const float2& x2, const float2& x3) {
                              a[0].x = fmaf(h, x0.x, a[0].x); a[0].y = fmaf(h, x0.y, a[0].y);
                              a[1].x = fmaf(h, x1.x, a[1].x); a[1].y = fmaf(h, x1.y, a[1].y);
                              a[2].x = fmaf(h, x2.x, a[2].x); a[2].y = fmaf(h, x2.y, a[2].y);
                              a[3].x = fmaf(h, x3.x, a[3].x); a[3].y = fmaf(h, x3.y, a[3].y);
                          });
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool neg = n0t - (hb + hs) + 4 * tid + r < 0;
            ft[r * FF_PF + tid + 1] = neg ? make_float2(0.f, 0.f) : a[r];
        }
    }
    __syncthreads();
    if (4 * tid < nd) {   // stage 2: u/l[j] = sum up/lo[k] f[j + hb - k]; d = rail(|u| / |l|) - 1
        // conjugate tap pair (see k_disc_2fsk): A = sum a x, B = sum b x with the real and imaginary parts of the UPPER filter's taps
        float ar[4] = {0.f, 0.f, 0.f, 0.f}, ai[4] = {0.f, 0.f, 0.f, 0.f}, br[4] = {0.f, 0.f, 0.f, 0.f}, bi[4] = {0.f, 0.f, 0.f, 0.f};
        const int nq = NQB > 0 ? NQB : (hb >> 2) + 1;
        int w = tid + (hb >> 2) + 1;
        float2 cur[4], nxt[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) cur[s] = ft[s * FF_PF + w];
        auto step = [&](const float2 h, const float2& x0, const float2& x1, const float2& x2, const float2& x3) {
            const float2 xs[4] = {x0, x1, x2, x3};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ar[r] = fmaf(h.x, xs[r].x, ar[r]); ai[r] = fmaf(h.x, xs[r].y, ai[r]);
                br[r] = fmaf(h.y, xs[r].x, br[r]); bi[r] = fmaf(h.y, xs[r].y, bi[r]);
            }
        };
        auto quad = [&](int q) {
            --w;
#pragma unroll
            for (int s = 0; s < 4; ++s) nxt[s] = ft[s * FF_PF + w];
            step(P.up[4 * q], cur[0], cur[1], cur[2], cur[3]);
            step(P.up[4 * q + 1], nxt[3], cur[0], cur[1], cur[2]);
            step(P.up[4 * q + 2], nxt[2], nxt[3], cur[0], cur[1]);
            step(P.up[4 * q + 3], nxt[1], nxt[2], nxt[3], cur[0]);
#pragma unroll
            for (int s = 0; s < 4; ++s) cur[s] = nxt[s];
        };
        if constexpr (NQB > 0) {
#pragma unroll 4
            for (int q = 0; q < NQB; ++q) quad(q);
        } else {
            for (int q = 0; q < nq; ++q) quad(q);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ur = ar[r] - bi[r], ui = ai[r] + br[r], lr = ar[r] + bi[r], li = ai[r] - br[r];
            const float mu = sqrtf(ur * ur + ui * ui);
            const float ml = sqrtf(lr * lr + li * li);
            float v = mu / ml;
            if (!(v >= 0.0f)) v = 0.0f;   // rail_ff lower bound; NaN (0/0) -> 0
            if (v > 2.0f) v = 2.0f;
            dt[r * FF_PD + tid + 1] = (n0t - hs + 4 * tid + r < 0) ? 0.f : v + (-1.0f);
        }
    }
    __syncthreads();
    if (4 * tid < T) {   // stage 3: y[o] = sum ts[k] d[o + hs - k]
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        ff_window4<FF_PD, NQS>(dt, tid + (hs >> 2) + 1, (hs >> 2) + 1, P.ts,
                          [&](float h, float x0, float x1, float x2, float x3) {
                              a[0] = fmaf(h, x0, a[0]); a[1] = fmaf(h, x1, a[1]); a[2] = fmaf(h, x2, a[2]); a[3] = fmaf(h, x3, a[3]);
                          });
        const uint32_t t0 = blockIdx.x * (uint32_t)T + 4u * tid;   // output index inside this call
        if (t0 == 0 && P.counts) P.counts[b * 4 + 0] = P.count;
        const int wf = tid + ((hb + hs) >> 2) + 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t t = t0 + r;
            if (t < P.count) {
                const int64_t n = n0t + 4 * tid + r;
                P.out.p[(size_t)b * (P.out.mask + 1u) + ((uint32_t)n & P.out.mask)] = a[r];
                if (P.port && t < P.port_cap) P.port[(size_t)b * P.port_cap + t] = ft[r * FF_PF + wf];
            }
        }
    }
}
// padded tap count of the fused kernel: 4 A + 1 >= n (tables hold 4 (A + 1) entries, zero filled)
int fsk2_ff_padded(int n) { return (n - 1 + 3) / 4 * 4 + 1; }
void launch_2fsk_ff(const Fsk2FfParams& p, int batch, hipStream_t s)
{
    if (!p.count) return;
    const int T = 1024 - (p.nb - 1) - (p.ns - 1);
    const dim3 grid((p.count + T - 1) / T, batch);
    // gr_demod_2fsk's designs at sps = 10 / 5 (2FSK-1k, -2k: 41 + 41 + 25 taps at 20 / 40 ksps) and at sps = 1 (10k: 80 ksps) are the instantiations
    static const bool rt = [] { const char* e = getenv("QRL_FF_RUNTIME_TAPS"); return e && atoi(e) != 0; }();
    const int qf = (p.nf - 1) / 4 + 1, qb = (p.nb - 1) / 4 + 1, qs = (p.ns - 1) / 4 + 1;
    if (!rt && qf == 11 && qb == 11 && qs == 7) hipLaunchKernelGGL((k_2fsk_ff<11, 11, 7>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((k_2fsk_ff<0, 0, 0>), grid, dim3(256), 0, s, p);
}

}  // namespace qrl
