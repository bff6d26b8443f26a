// kernels_decim_pl.hip — register-resident "phase-lane" decimating FIR (gfx950 / CDNA4).
//
//  k_decim_pl     : rotator_cc + rational_resampler_ccf(1, D, taps) on the caller's IQ, 32 < D <= 64, <= 16 taps per phase
//                   [gr_demod_base.cpp:57,180 (rotator); gr_demod_2fsk.cpp:82-88, gr_demod_gmsk.cpp:80-83,
//                    gr_demod_4fsk.cpp:86-91, gr_demod_bpsk.cpp:53-57 (the 1:50 first stage, 419 taps)]
//  k_decim_pl_gen : the same contract, one wave per output with checked fetches (call edges that need the carried
//                   history, and the second-stage form that reads an engine ring)
//
// Why not the matrix pipe here: the 1:50 stage has 8.4 real x complex MACs per input sample (33 flop / 8 B, a fifth of
// the f32 machine balance).  In the banded-Toeplitz MFMA form two thirds of the matrix work multiplies zero taps and the
// tile has to be staged through LDS; here NOTHING is staged:
//   * lane l of a wave owns polyphase branch p = D-1-l.  Block c of a stream = samples (c-1)D+1 .. cD: ONE coalesced
//     global_load_dwordx2 per wave and block (D x 8 contiguous bytes straight from HBM into a VGPR pair), 8 blocks in
//     flight per wave.
//   * the lane's J = ceil(nt/D) taps h[p + jD] live in registers.  The sample of block c is rotated (exact NCO tables in
//     LDS) and scattered into a ring of 16 running accumulators, acc[(c+j) & 15] += h[p+jD] * x (v_pk_fma_f32: re and im
//     at once).  After block c the accumulator of output m = c is complete in every lane: each sample is read once, no LDS
//     traffic for data, no barrier in the loop.
//   * 16 finished accumulators x 64 lanes are summed over the lanes by a TRANSPOSING butterfly (v_permlane32_swap,
//     v_permlane16_swap, DPP row rotations): every level halves the number of registers, 35 VALU per 16 outputs and
//     component instead of 6 x 16.  The tree is the radix-2 tree of the contract below.
// A wave walks a segment of S consecutive output blocks of one stream; the J-1 warm-up blocks in front of a segment are
// re-read (1.6 % at S = 512).  Outputs whose window reaches in front of this call's buffer go through k_decim_pl_gen.
//
// Summation contract "pl" (oracle/orc_blocks.c orc_decim_fir_ccf_pl): slot(i) = (i-1) mod D; per slot one chain, oldest
// sample first, first term a plain product, then fmaf; 64 slots (unused = +0) meet as v[l] += v[l+h], h = 32,16,...,1.
#include <vector>
#include "devmath.hpp"
#include "engine.hpp"

namespace qrl {

typedef float v2f __attribute__((ext_vector_type(2)));

#ifndef QRL_PL_PF
#define QRL_PL_PF 8
#endif
#ifndef QRL_PL_SCHED
#define QRL_PL_SCHED 0
#endif
#ifndef QRL_PL_SCALAR
#define QRL_PL_SCALAR 1   // plain v_fma_f32 pairs: packed f32 (v_pk_fma_f32) is no faster on gfx950 and measured 6 % slower here
#endif
#ifndef QRL_PL_WPE
#define QRL_PL_WPE 0
#endif
constexpr int PL_RING = 16;        // accumulator ring = unroll factor of the block loop
constexpr int PL_PF = QRL_PL_PF;   // blocks in flight per wave

__device__ __forceinline__ float pl_dpp_ror8(float v)
{
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x128 /* row_ror:8 */, 0xf, 0xf, true));
}
__device__ __forceinline__ float pl_dpp_xor4(float v)
{
    // lane ^ 4 inside a row: lanes of banks 0, 2 take l + 4 (row_shl:4), lanes of banks 1, 3 take l - 4 (row_shr:4)
    unsigned r = __builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x104 /* row_shl:4 */, 0xf, 0x5, false);
    r = __builtin_amdgcn_update_dpp(r, __float_as_uint(v), 0x114 /* row_shr:4 */, 0xf, 0xa, false);
    return __uint_as_float(r);
}
template <int CTRL>
__device__ __forceinline__ float pl_dpp_quad(float v)
{
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xf, 0xf, true));
}

// Sum 16 registers over the 64 lanes.  Returns, in every lane of quad q = lane >> 2, the lane sum of d[o(q)],
// o(q) = rowmap[q >> 2] + bankmap[q & 3], rowmap = {0, 2, 1, 3}, bankmap = {0, 8, 4, 12} (pl_out_index).
// Tree: v[l] + v[l+32], then +16, +8, +4, +2, +1.
__device__ __forceinline__ float pl_reduce16(const float (&d)[16], bool hi8, bool hi4)
{
    float b[8], c[4], e[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) {   // lanes 0-31: output 2i, lanes 32-63: output 2i + 1
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(d[2 * i]), __float_as_uint(d[2 * i + 1]), false, false);
        b[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // rows 0..3: outputs 4i, 4i + 2, 4i + 1, 4i + 3
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(b[2 * i]), __float_as_uint(b[2 * i + 1]), false, false);
        c[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {   // lanes 0-7 of a row: c[2i], lanes 8-15: c[2i + 1]
        const float keep = hi8 ? c[2 * i + 1] : c[2 * i], send = hi8 ? c[2 * i] : c[2 * i + 1];
        e[i] = keep + pl_dpp_ror8(send);
    }
    const float keep = hi4 ? e[1] : e[0], send = hi4 ? e[0] : e[1];   // banks 0, 2: e[0]; banks 1, 3: e[1]
    float f = keep + pl_dpp_xor4(send);
    f = f + pl_dpp_quad<0x4e>(f) /* quad_perm [2,3,0,1] */;
    f = f + pl_dpp_quad<0xb1>(f) /* quad_perm [1,0,3,2] */;
    return f;
}
__device__ __forceinline__ int pl_out_index(int lane)
{
    const int row = lane >> 4, bank = (lane >> 2) & 3;
    return ((row & 1) << 1 | (row >> 1)) + ((bank & 1) << 3 | (bank >> 1) << 2);
}

template <int J>
__global__ __launch_bounds__(256)
#if QRL_PL_WPE
__attribute__((amdgpu_waves_per_eu(QRL_PL_WPE, QRL_PL_WPE)))
#endif
void k_decim_pl(const DecimParams P_)
{
    const DecimParams& P = P_;
    __shared__ float2 t_lo[512];
    __shared__ float2 t_hi_all[4][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    t_lo[tid] = P.rot_lo[tid];
    t_lo[tid + 256] = P.rot_lo[tid + 256];

    // unit = (segment, stream); the waves of a workgroup take neighbouring streams of the same segment
    const uint32_t unit = blockIdx.x * 4u + (uint32_t)wave;
    const uint32_t B = P.pl_batch;
    const uint32_t seg = unit / B, b = unit - seg * B;
    const bool active = seg < P.pl_nseg;
    const int D = P.D;
    const uint64_t ms = P.pl_m_begin + (uint64_t)seg * P.pl_S;
    const uint64_t me = ms + P.pl_S < P.pl_m_end ? ms + P.pl_S : P.pl_m_end;
    const uint64_t c_first = ms - (uint64_t)(J - 1);                 // block c = samples (c-1) D + 1 .. c D
    const uint64_t i_first = (c_first - 1) * (uint64_t)D + 1;        // >= n0: the launcher only hands over interior outputs
    const uint32_t kb0 = (uint32_t)((i_first - P.rot_nbase) >> 9);
    float2* t_hi = t_hi_all[wave];
    if (active) t_hi[lane] = sincos_turn(P.rot_acc + ((uint64_t)(kb0 + (uint32_t)lane) << 9) * P.rot_inc);
    __syncthreads();
    if (!active) return;

    float h[J];
#pragma unroll
    for (int j = 0; j < J; ++j) h[j] = P.pl_taps[j * 64 + lane];
    const int lo = lane < D ? lane : D - 1;                          // idle lanes re-read the last sample against zero taps
    const int nblk = (int)(me - ms) + J - 1;
    const float2* ub = P.in + (size_t)b * P.in_stride + (size_t)(i_first - P.n0);   // wave-uniform
    const uint32_t k0 = (uint32_t)(i_first - P.rot_nbase) - (kb0 << 9);            // < 512
    const uint32_t lo8 = (uint32_t)lo * 8u;
    const bool hi8 = lane & 8, hi4 = lane & 4;
    const bool leader = (lane & 3) == 0;
    const int oidx = pl_out_index(lane);
    float2* orow = P.out.p + ((size_t)b * (P.out_row_mul_m1 + 1u) + P.out_row_add) * (P.out.mask + 1u);

    v2f pf[PL_PF];
#pragma unroll
    for (int q = 0; q < PL_PF; ++q) {
        const int t = q < nblk ? q : nblk - 1;
        const float2 v = ub[(size_t)t * D + lo];
        pf[q] = v2f{v.x, v.y};
    }
#if QRL_PL_SCALAR
    float ar[PL_RING], ai[PL_RING];
#pragma unroll
    for (int s = 0; s < PL_RING; ++s) ar[s] = ai[s] = 0.f;
#else
    v2f acc[PL_RING];
#pragma unroll
    for (int s = 0; s < PL_RING; ++s) acc[s] = v2f{0.f, 0.f};
#endif

    constexpr int UB = PL_PF > PL_RING ? PL_PF : PL_RING;   // blocks per loop body (multiple of both rings)
    const int nsup = (nblk + UB - 1) / UB;
    for (int sup = 0; sup < nsup; ++sup) {
#pragma unroll
        for (int grp = 0; grp < UB / PL_RING; ++grp) {
            float dr[PL_RING], di[PL_RING];
#pragma unroll
            for (int i = 0; i < PL_RING; ++i) {
                const int u = grp * PL_RING + i;
                const int t = sup * UB + u;
                const v2f xr = pf[u % PL_PF];
                {   // keep PL_PF blocks in flight (past the end of the segment: harmless re-read of its last block)
                    const int tn = t + PL_PF < nblk ? t + PL_PF : nblk - 1;
                    const float2 v = ub[(size_t)tn * D + lo];
                    pf[u % PL_PF] = v2f{v.x, v.y};
                }
                // rotator: phasor of sample k = T_hi[k >> 9] (x) T_lo[k & 511], byte addressed
                const uint32_t kb8 = (k0 + (uint32_t)t * (uint32_t)D) * 8u + lo8;
                const float2 plo = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(t_lo) + (kb8 & 4095u));
                const float2 phi = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(t_hi) + ((kb8 >> 9) & ~7u));
                const float2 xs = cmul_fma(make_float2(xr.x, xr.y), cmul_fma(phi, plo));
                const v2f x = v2f{xs.x, xs.y};
                // scatter into the ring: output m = c + j takes tap h[p + j D]; its first term (j = J - 1) is a plain product
#if QRL_PL_SCALAR
#pragma unroll
                for (int j = 0; j < J - 1; ++j) {
                    ar[(i + j) % PL_RING] = fmaf(h[j], xs.x, ar[(i + j) % PL_RING]);
                    ai[(i + j) % PL_RING] = fmaf(h[j], xs.y, ai[(i + j) % PL_RING]);
                }
                ar[(i + J - 1) % PL_RING] = h[J - 1] * xs.x;
                ai[(i + J - 1) % PL_RING] = h[J - 1] * xs.y;
                dr[i] = ar[i]; di[i] = ai[i];
                (void)x;
#else
#pragma unroll
                for (int j = 0; j < J - 1; ++j) acc[(i + j) % PL_RING] = __builtin_elementwise_fma(v2f{h[j], h[j]}, x, acc[(i + j) % PL_RING]);
                acc[(i + J - 1) % PL_RING] = v2f{h[J - 1], h[J - 1]} * x;
                dr[i] = acc[i].x; di[i] = acc[i].y;
#endif
#if QRL_PL_SCHED
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
            const float yr = pl_reduce16(dr, hi8, hi4), yi = pl_reduce16(di, hi8, hi4);
            const uint64_t m = c_first + (uint64_t)(sup * UB + grp * PL_RING + oidx);
            if (leader && m >= ms && m < me) orow[(uint32_t)m & P.out.mask] = make_float2(yr, yi);
        }
    }
}

// One wave per output, checked fetches: zero in front of the stream, carried (already rotated) history in front of this
// call's buffer, the caller's buffer with the exact NCO, or an engine ring.
__global__ __launch_bounds__(256) void k_decim_pl_gen(const DecimParams P_, uint64_t m_first, uint32_t count)
{
    const DecimParams& P = P_;
    const int lane = threadIdx.x & 63;
    const uint32_t o = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (o >= count) return;
    const int b = blockIdx.y;
    const uint64_t m = m_first + o;
    const int D = P.D, J = P.pl_J;
    float vr = 0.f, vi = 0.f;
    if (lane < D) {
        for (int j = J - 1; j >= 0; --j) {
            const int k = (D - 1 - lane) + j * D;
            const float hk = k < P.nt ? P.pl_taps[j * 64 + lane] : 0.f;
            const int64_t i = (int64_t)m * D - k;
            float2 x = make_float2(0.f, 0.f);
            if (i >= 0) {
                const uint64_t ui = (uint64_t)i;
                if (P.in) {
                    if (ui >= P.n0) {
                        x = P.in[(size_t)b * P.in_stride + (size_t)(ui - P.n0)];
                        if (P.rot_enable) {
                            const uint64_t kk = ui - P.rot_nbase;
                            const float2 hi = sincos_turn(P.rot_acc + ((kk >> 9) << 9) * P.rot_inc);
                            x = cmul_fma(x, cmul_fma(hi, P.rot_lo[(uint32_t)kk & 511u]));
                        }
                    } else {
                        const uint64_t d = P.n0 - ui;
                        if (d <= P.hist_len) x = P.hist[(size_t)b * P.hist_len + (P.hist_len - (uint32_t)d)];
                    }
                } else {
                    x = P.in_ring.p[(size_t)b * (P.in_ring.mask + 1u) + ((uint32_t)ui & P.in_ring.mask)];
                }
            }
            if (j == J - 1) { vr = hk * x.x; vi = hk * x.y; }
            else { vr = fmaf(hk, x.x, vr); vi = fmaf(hk, x.y, vi); }
        }
    }
#pragma unroll
    for (int hh = 32; hh >= 1; hh >>= 1) {
        vr = vr + __shfl_xor(vr, hh);
        vi = vi + __shfl_xor(vi, hh);
    }
    if (lane == 0)
        P.out.p[((size_t)b * (P.out_row_mul_m1 + 1u) + P.out_row_add) * (P.out.mask + 1u) + ((uint32_t)m & P.out.mask)] = make_float2(vr, vi);
}

// rule shared with oracle/orc_blocks.c orc_decim_uses_pl
bool decim_uses_pl(int nt, int D) { return D > 32 && D <= 64 && (nt + D - 1) / D <= 16; }

// lane tap table [J][64]: lane l holds h[(j + 1) D - 1 - l]
std::vector<float> decim_pl_layout(const std::vector<float>& h, int D)
{
    const int nt = (int)h.size(), J = (nt + D - 1) / D;
    std::vector<float> t((size_t)J * 64, 0.0f);
    for (int j = 0; j < J; ++j)
        for (int l = 0; l < D; ++l) {
            const int k = (j + 1) * D - 1 - l;
            if (k < nt) t[(size_t)j * 64 + l] = h[k];
        }
    return t;
}

template <int J>
static void pl_launch_main(const DecimParams& q, uint32_t units, hipStream_t s)
{
    hipLaunchKernelGGL((k_decim_pl<J>), dim3((units + 3) / 4), dim3(256), 0, s, q);
}

int launch_decim_pl(const DecimParams& p, int batch, hipStream_t s)
{
    if (p.m_count == 0) return 0;
    const int D = p.D, J = (p.nt + D - 1) / D;
    DecimParams q = p;
    q.pl_J = J;
    const uint64_t m_end = p.m0 + p.m_count;
    uint64_t m_main = m_end;   // first output of the register kernel
    if (p.in && p.rot_enable) {
        // interior outputs: every sample the lanes touch, (m - J) D + 1 .. m D, lies in this call's buffer
        const uint64_t need = p.n0 + (uint64_t)J * D;            // m D >= n0 + J D - 1 + ... : m >= ceil((n0 - 1) / D) + J
        m_main = (need + D - 2) / D;                              // smallest m with (m - J) D + 1 >= n0
        if (m_main < p.m0) m_main = p.m0;
        if (m_main > m_end) m_main = m_end;
    }
    if (m_main > p.m0) {
        const uint32_t cnt = (uint32_t)(m_main - p.m0);
        hipLaunchKernelGGL(k_decim_pl_gen, dim3((cnt + 3) / 4, batch), dim3(256), 0, s, q, p.m0, cnt);
    }
    if (m_main >= m_end) return 0;
    // segment length: a multiple of 16 blocks, long enough to keep the warm-up re-reads small, short enough to spread the
    // call over >= ~16 waves per CU, and inside the 64-entry coarse rotator table of a wave (64 x 512 samples)
    const uint64_t total = (m_end - m_main) * (uint64_t)batch;
    uint64_t S = total / (256u * 16u * 4u);
    const uint64_t s_cap = (uint64_t)((62 * 512) / D - J) / 16 * 16;
    if (S > 512) S = 512;
    if (S > s_cap) S = s_cap;
    S = S / 16 * 16;
    if (S < 16) S = 16;
    q.pl_S = (uint32_t)S;
    q.pl_m_begin = m_main; q.pl_m_end = m_end;
    q.pl_nseg = (uint32_t)((m_end - m_main + S - 1) / S);
    q.pl_batch = (uint32_t)batch;
    const uint32_t units = q.pl_nseg * (uint32_t)batch;
    switch (J) {
    case 1: pl_launch_main<1>(q, units, s); break;   case 2: pl_launch_main<2>(q, units, s); break;
    case 3: pl_launch_main<3>(q, units, s); break;   case 4: pl_launch_main<4>(q, units, s); break;
    case 5: pl_launch_main<5>(q, units, s); break;   case 6: pl_launch_main<6>(q, units, s); break;
    case 7: pl_launch_main<7>(q, units, s); break;   case 8: pl_launch_main<8>(q, units, s); break;
    case 9: pl_launch_main<9>(q, units, s); break;   case 10: pl_launch_main<10>(q, units, s); break;
    case 11: pl_launch_main<11>(q, units, s); break; case 12: pl_launch_main<12>(q, units, s); break;
    case 13: pl_launch_main<13>(q, units, s); break; case 14: pl_launch_main<14>(q, units, s); break;
    case 15: pl_launch_main<15>(q, units, s); break; default: pl_launch_main<16>(q, units, s); break;
    }
    return 0;
}

}  // namespace qrl
