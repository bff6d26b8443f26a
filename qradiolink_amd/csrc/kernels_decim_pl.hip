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
//     LDS) and scattered into a ring of 16 running accumulators, acc[(c+j) & 15] += h[p+jD] * x (plain v_fma_f32 pairs:
//     this kernel is HBM bound, the packed form measured 6 % slower).  After block c the accumulator of output m = c is complete in every lane: each sample is read once, no LDS
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

constexpr int PL_RING = 16;        // accumulator ring = unroll factor of the block loop
constexpr int PL_PF = 8;           // blocks in flight per wave

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
// the levels of pl_reduce16 one by one (k_decim_plx feeds the tree as outputs finish)
__device__ __forceinline__ float pl_level32(float d0, float d1)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(d0), __float_as_uint(d1), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float pl_level16(float b0, float b1)
{
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(b0), __float_as_uint(b1), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float pl_level8(float c0, float c1, bool hi8)
{
    const float keep = hi8 ? c1 : c0, send = hi8 ? c0 : c1;
    return keep + pl_dpp_ror8(send);
}
__device__ __forceinline__ float pl_level421(float e0, float e1, bool hi4)
{
    const float keep = hi4 ? e1 : e0, send = hi4 ? e0 : e1;
    float f = keep + pl_dpp_xor4(send);
    f = f + pl_dpp_quad<0x4e>(f);
    f = f + pl_dpp_quad<0xb1>(f);
    return f;
}
__device__ __forceinline__ int pl_out_index(int lane)
{
    const int row = lane >> 4, bank = (lane >> 2) & 3;
    return ((row & 1) << 1 | (row >> 1)) + ((bank & 1) << 3 | (bank >> 1) << 2);
}

template <int J>
__global__ __launch_bounds__(256)
void k_decim_pl(const DecimParams P_)
{
    const DecimParams& P = P_;
    __shared__ float2 t_lo[512];
    __shared__ float2 t_one[512];
    __shared__ float2 t_hi_all[4][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    t_lo[tid] = P.rot_lo[tid];
    t_lo[tid + 256] = P.rot_lo[tid + 256];
    t_one[tid] = make_float2(1.f, 0.f);
    t_one[tid + 256] = make_float2(1.f, 0.f);

    // unit = (stream, segment); the waves of a workgroup take neighbouring segments of one stream.  Behind the regular units:
    // one EDGE unit per stream (outputs pl_edge_ms .. pl_edge_me out of the staged, already rotated scratch: identity phasors)
    const uint32_t unit = blockIdx.x * 4u + (uint32_t)wave;
    const uint32_t B = P.pl_batch;
    const uint32_t nreg = P.pl_nseg * B;
    const bool edge = unit >= nreg;
    // consecutive units = consecutive segments of ONE stream: the resident waves then sweep a contiguous ~1 GB region instead of
    // one 200 KB piece out of each of 5000 rows 2 MB apart (measured, C1: 6.85 ms on every run against 6.9 - 8.3 ms depending on
    // where the process's input buffer happened to be mapped)
    const uint32_t nsg = P.pl_nseg ? P.pl_nseg : 1u;
    const uint32_t b = edge ? unit - nreg : unit / nsg, seg = edge ? 0u : unit - b * nsg;
    const bool active = edge ? (b < B && P.pl_edge_me > P.pl_edge_ms) : true;
    const int D = P.D;
    const uint64_t ms = edge ? P.pl_edge_ms : P.pl_m_begin + (uint64_t)seg * P.pl_S;
    const uint64_t me = edge ? P.pl_edge_me : (ms + P.pl_S < P.pl_m_end ? ms + P.pl_S : P.pl_m_end);
    // block c = samples (c-1) D + 1 .. c D; the first block of the unit is ms - (J - 1) (edge units: may lie in front of the stream)
    const int64_t c_first_s = (int64_t)ms - (int64_t)(J - 1);
    const uint64_t c_first = (uint64_t)c_first_s;
    const int64_t i_first_s = (c_first_s - 1) * (int64_t)D + 1;
    const uint64_t i_first = (uint64_t)i_first_s;                    // regular units: >= n0 (the launcher only hands over interior outputs)
    const uint32_t kb0 = edge ? 0u : (uint32_t)((i_first - P.rot_nbase) >> 9);
    float2* t_hi = t_hi_all[wave];
    if (active) t_hi[lane] = edge ? make_float2(1.f, 0.f) : sincos_turn(P.rot_acc + ((uint64_t)(kb0 + (uint32_t)lane) << 9) * P.rot_inc);
    const float2* tl = edge ? t_one : t_lo;
    __syncthreads();
    if (!active) return;

    float h[J];
#pragma unroll
    for (int j = 0; j < J; ++j) h[j] = P.pl_taps[j * 64 + lane];
    const int lo = lane < D ? lane : D - 1;                          // idle lanes re-read the last sample against zero taps
    const int nblk = (int)(me - ms) + J - 1;
    const float2* ub = edge ? P.pl_edge + (size_t)b * P.pl_edge_stride : P.in + (size_t)b * P.in_stride + (size_t)(i_first - P.n0);   // wave-uniform
    const uint32_t k0 = edge ? 0u : (uint32_t)(i_first - P.rot_nbase) - (kb0 << 9);   // < 512
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
    float ar[PL_RING], ai[PL_RING];
#pragma unroll
    for (int s = 0; s < PL_RING; ++s) ar[s] = ai[s] = 0.f;

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
                const float2 plo = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(tl) + (kb8 & 4095u));
                const float2 phi = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(t_hi) + ((kb8 >> 9) & ~7u));
                const float2 xs = cmul_fma(make_float2(xr.x, xr.y), cmul_fma(phi, plo));
                // scatter into the ring: output m = c + j takes tap h[p + j D]; its first term (j = J - 1) is a plain product
#pragma unroll
                for (int j = 0; j < J - 1; ++j) {
                    ar[(i + j) % PL_RING] = fmaf(h[j], xs.x, ar[(i + j) % PL_RING]);
                    ai[(i + j) % PL_RING] = fmaf(h[j], xs.y, ai[(i + j) % PL_RING]);
                }
                ar[(i + J - 1) % PL_RING] = h[J - 1] * xs.x;
                ai[(i + J - 1) % PL_RING] = h[J - 1] * xs.y;
                dr[i] = ar[i]; di[i] = ai[i];
            }
            const float yr = pl_reduce16(dr, hi8, hi4), yi = pl_reduce16(di, hi8, hi4);
            const uint64_t m = c_first + (uint64_t)(sup * UB + grp * PL_RING + oidx);
            if (leader && m >= ms && m < me) orow[(uint32_t)m & P.out.mask] = make_float2(yr, yi);
        }
    }
}

// ---- k_decim_pl2: the same kernel with the input routed HBM -> LDS by LDS-DMA ------------------------------------------------------
// k_decim_pl reads each 400-byte block with one global_load_dwordx2 per wave: 50 lanes x 8 bytes, four or five 128-byte lines touched for
// 3.1 lines of data, partial lines requested twice.  That request stream -- not HBM -- capped it at 5.1 TB/s (tools/ubench/
// stream_patterns: 5.27 TB/s for exactly this pattern).  Measured with tools/ubench/stream_lds.hip (round 3): a wave that streams
// its segment as 1 KiB LDS-DMA pieces (global_load_lds_dwordx4, 64 lanes x 16 bytes, whole lines) into a PRIVATE ring of 8 KiB with
// 4 pieces in flight, non-temporal policy, reaches 7.07 TB/s at 16-20 waves per CU -- with 20 FMAs per sample beside it.
//   * ring = 8 pieces of 1 KiB per wave; piece q of the wave's byte stream (a0 + 1024 q, a0 = segment start rounded down to 128 bytes
//     relative to the stream's row) lands at ring offset (1024 q) mod 8192.  No barrier anywhere: the wave that issued a piece is the
//     only one that reads it, ordered by its own counted s_waitcnt vmcnt.
//   * every group of 4 blocks (1600 bytes) tops the DMA queue up to need + 4 pieces, need = pieces that cover the group; then
//     s_waitcnt vmcnt(4): at most the 4 pieces YOUNGER than the last needed one are outstanding (stores in between only make the
//     wait stricter), so everything the group reads has landed.  The ring never holds more than 4 + 3 live pieces.
//   * lane l reads its sample of block t with one ds_read_b64 at (o0 + 400 t + 8 l) mod 8192: lane-contiguous, conflict free.
// Everything behind the sample fetch -- rotator, tap scatter, accumulator ring, transposing reduction -- is k_decim_pl's: the "pl"
// summation contract is untouched, results are bit-identical.
#ifndef QRL_PL2_PD
#define QRL_PL2_PD 6
#endif
#ifndef QRL_PL2_G
#define QRL_PL2_G 2
#endif
#ifndef QRL_PL2_RP
#define QRL_PL2_RP 8
#endif
constexpr int PL2_RP = QRL_PL2_RP;     // ring pieces (1 KiB each) per wave (power of two): the ring is aligned to its size in LDS
constexpr int PL2_PD = QRL_PL2_PD;     // pieces in flight behind the last needed one
constexpr int PL2_G = QRL_PL2_G;       // blocks per DMA top-up / wait group (divides PL_RING)
// a group of G blocks of <= 512 bytes spans at most ceil((1023 + 512 G) / 1024) pieces; the ring holds them + the PD pieces in flight
static_assert((PL2_RP & (PL2_RP - 1)) == 0 && PL_RING % PL2_G == 0 && PL2_PD + (1023 + 512 * PL2_G + 1023) / 1024 <= PL2_RP, "ring too small for the group / depth");

__device__ __forceinline__ void pl2_glds16(const void* gsrc, uint32_t lds_dst)
{
    // one LDS-DMA piece, non-temporal: 64 lanes x 16 B from per-lane global addresses to LDS[lds_dst + 16 lane].  M0 (compiler
    // reserved) is saved and restored inside the statement (cdna_hip_programming.md, "LDS-DMA recipe")
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
typedef const __attribute__((address_space(3))) v2f* pl2_lds_f2;
__device__ __forceinline__ float2 pl2_lds(uint32_t addr)   // ds_read_b64 from a raw LDS byte address
{
    const v2f v = *(pl2_lds_f2)(uintptr_t)addr;
    return make_float2(v.x, v.y);
}

__device__ __forceinline__ uint32_t pl2_lds_addr(const void* p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p; }

template <int J>
__global__ __launch_bounds__(256)
void k_decim_pl2(const DecimParams P_)
{
    const DecimParams& P = P_;
    __shared__ __align__(PL2_RP * 1024) unsigned char ring_all[4 * PL2_RP * 1024];   // ring of wave w at LDS byte w * ring size (+ a multiple of it)
    __shared__ float2 t_lo[512];            // fine rotator table; entry 0 is exactly (1, 0): what edge units read through index mask 0
    __shared__ float2 t_hi_all[4][64];      // coarse rotator table of each wave's segment
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    t_lo[tid] = P.rot_lo[tid];
    t_lo[tid + 256] = P.rot_lo[tid + 256];

    const uint32_t unit = blockIdx.x * 4u + (uint32_t)wave;
    const uint32_t B = P.pl_batch;
    const uint32_t nreg = P.pl_nseg * B;
    const bool edge = unit >= nreg;
    const uint32_t nsg = P.pl_nseg ? P.pl_nseg : 1u;
    const uint32_t b = edge ? unit - nreg : unit / nsg, seg = edge ? 0u : unit - b * nsg;
    const bool active = edge ? (b < B && P.pl_edge_me > P.pl_edge_ms) : true;
    const int D = P.D;
    const uint64_t ms = edge ? P.pl_edge_ms : P.pl_m_begin + (uint64_t)seg * P.pl_S;
    const uint64_t me = edge ? P.pl_edge_me : (ms + P.pl_S < P.pl_m_end ? ms + P.pl_S : P.pl_m_end);
    const int64_t c_first_s = (int64_t)ms - (int64_t)(J - 1);
    const uint64_t c_first = (uint64_t)c_first_s;
    const int64_t i_first_s = (c_first_s - 1) * (int64_t)D + 1;
    const uint64_t i_first = (uint64_t)i_first_s;
    const uint32_t kb0 = edge ? 0u : (uint32_t)((i_first - P.rot_nbase) >> 9);
    if (active) t_hi_all[wave][lane] = edge ? make_float2(1.f, 0.f) : sincos_turn(P.rot_acc + ((uint64_t)(kb0 + (uint32_t)lane) << 9) * P.rot_inc);
    __syncthreads();
    if (!active) return;

    float h[J];
#pragma unroll
    for (int j = 0; j < J; ++j) h[j] = P.pl_taps[j * 64 + lane];
    const int lo = lane < D ? lane : D - 1;                          // idle lanes re-read the last sample against zero taps
    const int nblk = (int)(me - ms) + J - 1;
    // the wave's byte stream: row = the stream's buffer (or its edge scratch), first byte of block 0 at row + off0
    const unsigned char* rowp = reinterpret_cast<const unsigned char*>(edge ? P.pl_edge + (size_t)b * P.pl_edge_stride : P.in + (size_t)b * P.in_stride);
    const uint64_t off0 = edge ? 0ull : (uint64_t)(i_first - P.n0) * 8ull;
    const uint64_t row_bytes = edge ? (uint64_t)P.pl_edge_stride * 8ull : (uint64_t)P.n * 8ull;   // multiples of 16 (even sample counts)
    const uint32_t o0 = (uint32_t)(off0 & 127u);                     // offset of block 0 inside piece 0
    const uint64_t a_off = off0 - o0;                                // piece 0 starts here (128-byte aligned relative to the row)
    const uint32_t bpb = (uint32_t)D * 8u;                           // bytes per block
    const uint32_t npieces = (o0 + (uint32_t)nblk * bpb + 1023u) >> 10;
    // pieces [0, q_safe) lie inside the row; the lanes of later pieces are clamped to the row's last 16 bytes (never consumed)
    const uint32_t q_safe = (uint32_t)((row_bytes - a_off) >> 10);
    const uint32_t rbase = pl2_lds_addr(ring_all) + (uint32_t)wave * (PL2_RP * 1024u);   // multiple of the ring size
    const uint32_t tlo_base = pl2_lds_addr(t_lo);
    const uint32_t thi_base = pl2_lds_addr(t_hi_all) + (uint32_t)wave * 512u;
    const unsigned char* gp = rowp + a_off + (size_t)lane * 16;      // this lane's 16 bytes of the next piece
    const unsigned char* last16 = rowp + row_bytes - 16;
    uint32_t issued = 0;
    auto issue_upto = [&](uint32_t want) {                           // wave uniform
        while (issued < want) {
            const uint32_t dst = rbase + (issued & (PL2_RP - 1)) * 1024u;
            if (issued < q_safe) pl2_glds16(gp, dst);
            else pl2_glds16(gp < last16 ? gp : last16, dst);
            gp += 1024;
            ++issued;
        }
    };
    const uint32_t k0 = edge ? 0u : (uint32_t)(i_first - P.rot_nbase) - (kb0 << 9);   // < 512
    const uint32_t lo8 = (uint32_t)lo * 8u;
    const uint32_t tl_mask = edge ? 0u : 4095u;
    uint32_t xa = rbase | ((o0 + lo8) & (PL2_RP * 1024u - 1u));     // LDS address of this lane's sample of the current block
    uint32_t kb8 = k0 * 8u + lo8;                                    // 8 x (NCO index of that sample, relative to the coarse table)
    const bool hi8 = lane & 8, hi4 = lane & 4;
    const bool leader = (lane & 3) == 0;
    const int oidx = pl_out_index(lane);
    float2* orow = P.out.p + ((size_t)b * (P.out_row_mul_m1 + 1u) + P.out_row_add) * (P.out.mask + 1u);

    float ar[PL_RING], ai[PL_RING];
#pragma unroll
    for (int s = 0; s < PL_RING; ++s) ar[s] = ai[s] = 0.f;

    const int nsup = (nblk + PL_RING - 1) / PL_RING;
    uint32_t need_bytes = o0 + 1023u;                                // (bytes up to the end of the current group) + 1023
    for (int sup = 0; sup < nsup; ++sup) {
        float dr[PL_RING], di[PL_RING];
#pragma unroll
        for (int grp = 0; grp < PL_RING / PL2_G; ++grp) {
            {   // pieces that cover the blocks of this group (clamped to the segment), + PL2_PD in flight behind them
                need_bytes += (uint32_t)PL2_G * bpb;
                uint32_t need = need_bytes >> 10;
                need = need < npieces ? need : npieces;
                const uint32_t want = need + PL2_PD < npieces ? need + PL2_PD : npieces;   // never past the segment's last piece
                // the slots refilled now were last read a group ago: those ds_reads have been issued (program order, "memory"
                // clobber) -- make sure they have also RETURNED before a DMA can overwrite them
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                issue_upto(want);
                if (want - need == PL2_PD) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PL2_PD) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // the last groups of a segment
            }
#pragma unroll
            for (int ii = 0; ii < PL2_G; ++ii) {
                const int i = grp * PL2_G + ii;
                const float2 xr = pl2_lds(xa);
                // rotator: phasor of sample k = T_hi[k >> 9] (x) T_lo[k & 511], byte addressed
                const float2 plo = pl2_lds(tlo_base + (kb8 & tl_mask));
                const float2 phi = pl2_lds(thi_base + ((kb8 >> 12) << 3));
                xa = ((xa + bpb) & (PL2_RP * 1024u - 1u)) | rbase;
                kb8 += bpb;
                const float2 xs = cmul_fma(xr, cmul_fma(phi, plo));
#pragma unroll
                for (int j = 0; j < J - 1; ++j) {
                    ar[(i + j) % PL_RING] = fmaf(h[j], xs.x, ar[(i + j) % PL_RING]);
                    ai[(i + j) % PL_RING] = fmaf(h[j], xs.y, ai[(i + j) % PL_RING]);
                }
                ar[(i + J - 1) % PL_RING] = h[J - 1] * xs.x;
                ai[(i + J - 1) % PL_RING] = h[J - 1] * xs.y;
                dr[i] = ar[i]; di[i] = ai[i];
            }
        }
        const float yr = pl_reduce16(dr, hi8, hi4), yi = pl_reduce16(di, hi8, hi4);
        const uint64_t m = c_first + (uint64_t)(sup * PL_RING + oidx);
        if (leader && m >= ms && m < me) orow[(uint32_t)m & P.out.mask] = make_float2(yr, yi);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no DMA may still be writing this wave's ring when the workgroup's LDS is handed on
}

// ---- generalised geometry: E samples per lane and block, R outputs per block ------------------------------------------------------
//  k_decim_plx<E, R, U>: the same register-resident scheme for the front ends whose decimation is not one sample per lane:
//    E = 2, R = 1   64 < D <= 128 (the 100:1 front end of a 100 Msps device, 4 181 taps): a block is D samples, lane l owns the
//                   samples 2l and 2l + 1 of every block (slot(i) = ((i - 1) mod D) / 2 of the "pl" contract);
//    (E = 1, R = 2, the 25:1 front end as blocks of 2 D samples, was measured too: 2.47 ms on C2 against 1.9 ms for the MFMA
//    kernel -- one sample per lane and block pays the per-block overhead twice; the rule keeps 25:1 on the m16 contract.)
//  Block c = samples (c-1) D' + 1 .. c D' (D' = R D).  The sample r = E l + e of block c meets output m = (c-1) R + u with tap
//  h[u D - 1 - r], u = 1 .. U = floor((nt + D' - 1) / D); the accumulators are a sliding window of 16 - R + U registers that moves
//  down by 16 after every 16 outputs.  With 84 taps + 114 accumulators + the prefetch ring a wave needs > 256 VGPRs, so one wave
//  runs per SIMD: a lone wave issues one VALU per ~4-5 cycles, and v_pk_fma_f32 (re and im in one instruction) is what keeps the
//  f32 pipe busy at that rate (32 lanes per clock: two v_fma_f32 = one v_pk_fma_f32 = 4 cycles).
constexpr int PLX_PF = 4;          // blocks in flight per wave (two waves per SIMD: 247 VGPRs at 4, 276 at 8)
constexpr int PLX_NHI = 256;       // coarse rotator entries per wave: a segment spans <= 255 x 512 samples

// cmul_fma (devmath.hpp) on register pairs: {fma(a.x, b.x, -(a.y b.y)), fma(a.x, b.y, a.y b.x)} = two packed instructions
__device__ __forceinline__ v2f plx_cmul_fma(v2f a, v2f b)
{
    const v2f t = v2f{-a.y, a.y} * v2f{b.y, b.x};   // -(a.y b.y) == (-a.y) b.y exactly
    return __builtin_elementwise_fma(v2f{a.x, a.x}, b, t);
}

// acc += {h, h} * x with h = one half of a tap pair, broadcast by op_sel.  (Written as a v2f splat the compiler hoists the
// loop-invariant {h, h} pairs out of the loop: two registers per tap, the taps alone would not fit the VGPR file.)
template <int HI>
__device__ __forceinline__ void plx_fma(v2f& acc, v2f hpair, v2f x)
{
    if (HI) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(hpair), "v"(x));
    else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(hpair), "v"(x));
}
template <int HI>
__device__ __forceinline__ v2f plx_mul(v2f hpair, v2f x)
{
    v2f r;
    if (HI) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "v"(hpair), "v"(x));
    else asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(hpair), "v"(x));
    return r;
}

template <int E, int R, int U>
__global__ __launch_bounds__(256, 2) void k_decim_plx(const DecimParams P_)
{
    constexpr int G = 16 / R;              // blocks per group of 16 outputs
    constexpr int WN = 16 - R + U;         // accumulator window
    constexpr int WU = (U - 1) / R;        // warm-up blocks in front of a segment
    static_assert(G % PLX_PF == 0 || PLX_PF % G == 0, "prefetch ring and group must nest");
    const DecimParams& P = P_;
    __shared__ float2 t_lo[512];
    __shared__ float2 t_one[512];
    __shared__ float2 t_hi_all[4][PLX_NHI];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    t_lo[tid] = P.rot_lo[tid];
    t_lo[tid + 256] = P.rot_lo[tid + 256];
    t_one[tid] = make_float2(1.f, 0.f);
    t_one[tid + 256] = make_float2(1.f, 0.f);

    const uint32_t unit = blockIdx.x * 4u + (uint32_t)wave;
    const uint32_t B = P.pl_batch;
    const uint32_t nreg = P.pl_nseg * B;
    const bool edge = unit >= nreg;                                    // one edge unit per stream behind the regular ones (see k_decim_pl)
    // consecutive units = consecutive segments of ONE stream: the resident waves then sweep a contiguous ~1 GB region instead of
    // one 200 KB piece out of each of 5000 rows 2 MB apart (measured, C1: 6.85 ms on every run against 6.9 - 8.3 ms depending on
    // where the process's input buffer happened to be mapped)
    const uint32_t nsg = P.pl_nseg ? P.pl_nseg : 1u;
    const uint32_t b = edge ? unit - nreg : unit / nsg, seg = edge ? 0u : unit - b * nsg;
    const bool active = edge ? (b < B && P.pl_edge_me > P.pl_edge_ms) : true;
    const int Dp = P.D * R, NL = Dp / E;
    const uint64_t ms = edge ? P.pl_edge_ms : P.pl_m_begin + (uint64_t)seg * P.pl_S;   // == 1 (mod R)
    const uint64_t me = edge ? P.pl_edge_me : (ms + P.pl_S < P.pl_m_end ? ms + P.pl_S : P.pl_m_end);
    const int64_t c_first_s = (int64_t)((ms - 1) / R) + 1 - (int64_t)WU;
    const uint64_t c_first = (uint64_t)c_first_s;
    const uint64_t i_first = (uint64_t)((c_first_s - 1) * (int64_t)Dp + 1);   // regular units: >= n0 (interior outputs only)
    const uint32_t kb0 = edge ? 0u : (uint32_t)((i_first - P.rot_nbase) >> 9);
    float2* t_hi = t_hi_all[wave];
    if (active) {
#pragma unroll
        for (int q = 0; q < PLX_NHI / 64; ++q)
            t_hi[lane + 64 * q] = edge ? make_float2(1.f, 0.f) : sincos_turn(P.rot_acc + ((uint64_t)(kb0 + (uint32_t)(lane + 64 * q)) << 9) * P.rot_inc);
    }
    const float2* tl = edge ? t_one : t_lo;
    __syncthreads();
    if (!active) return;

    // taps two to a register pair: v_pk_fma_f32 broadcasts either half through op_sel, a scalar float operand would be
    // widened to a {h, h} pair by the compiler (twice the registers)
    constexpr int NH = (E * U + 1) / 2;
    v2f hp[NH];
#pragma unroll
    for (int q = 0; q < NH; ++q) {
        hp[q].x = P.pl_taps[(2 * q) * 64 + lane];
        hp[q].y = 2 * q + 1 < E * U ? P.pl_taps[(2 * q + 1) * 64 + lane] : 0.f;
    }
    const int lo = lane < NL ? lane : NL - 1;                         // idle lanes re-read the last samples against zero taps
    const int nblk = WU + (int)((me - ms + R - 1) / R);
    const float2* ub = (edge ? P.pl_edge + (size_t)b * P.pl_edge_stride : P.in + (size_t)b * P.in_stride + (size_t)(i_first - P.n0)) + (size_t)(E * lo);   // lane's first sample of block 0
    const uint32_t k0 = (edge ? 0u : (uint32_t)(i_first - P.rot_nbase) - (kb0 << 9)) + (uint32_t)(E * lo);   // its NCO index, relative
    const bool hi8 = lane & 8, hi4 = lane & 4;
    const bool leader = (lane & 3) == 0;
    const int oidx = pl_out_index(lane);
    float2* orow = P.out.p + ((size_t)b * (P.out_row_mul_m1 + 1u) + P.out_row_add) * (P.out.mask + 1u);

    v2f pf[PLX_PF][E];
#pragma unroll
    for (int q = 0; q < PLX_PF; ++q) {
        const int t = q < nblk ? q : nblk - 1;
#pragma unroll
        for (int e = 0; e < E; ++e) { const float2 v = ub[(size_t)t * Dp + e]; pf[q][e] = v2f{v.x, v.y}; }
    }
    v2f acc[WN];
#pragma unroll
    for (int a = 0; a < WN; ++a) acc[a] = v2f{0.f, 0.f};

    const int ngrp = (nblk + G - 1) / G;
    for (int grp = 0; grp < ngrp; ++grp) {
        float dr[16], di[16];
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int t = grp * G + i;
            v2f x[E];
#pragma unroll
            for (int e = 0; e < E; ++e) x[e] = pf[i % PLX_PF][e];
            {   // keep PLX_PF blocks in flight (past the end of the segment: harmless re-read of its last block)
                const int tn = t + PLX_PF < nblk ? t + PLX_PF : nblk - 1;
#pragma unroll
                for (int e = 0; e < E; ++e) { const float2 v = ub[(size_t)tn * Dp + e]; pf[i % PLX_PF][e] = v2f{v.x, v.y}; }
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {   // rotator: phasor of sample k = T_hi[k >> 9] (x) T_lo[k & 511]
                const uint32_t kb8 = (k0 + (uint32_t)t * (uint32_t)Dp + (uint32_t)e) * 8u;
                const float2 plo = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(tl) + (kb8 & 4095u));
                const float2 phi = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(t_hi) + ((kb8 >> 9) & ~7u));
                x[e] = plx_cmul_fma(x[e], plx_cmul_fma(v2f{phi.x, phi.y}, v2f{plo.x, plo.y}));
            }
            // output (c-1) R + u sits at window slot i R + u - 1; its chain starts (plain product) at its oldest block, u > U - R
#pragma unroll
            for (int u = U; u >= 1; --u) {
                const int a = i * R + u - 1;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int qq = (u - 1) * E + e;
                    if (u > U - R && e == 0) acc[a] = (qq & 1) ? plx_mul<1>(hp[qq >> 1], x[e]) : plx_mul<0>(hp[qq >> 1], x[e]);
                    else if (qq & 1) plx_fma<1>(acc[a], hp[qq >> 1], x[e]);
                    else plx_fma<0>(acc[a], hp[qq >> 1], x[e]);
                }
            }
            // finished outputs enter the transposing tree at once (pl_reduce16 level by level: fewer live registers)
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const int o = i * R + q;
                dr[o] = acc[o].x; di[o] = acc[o].y;
                if ((o & 1) == 1) { dr[o >> 1] = pl_level32(dr[o - 1], dr[o]); di[o >> 1] = pl_level32(di[o - 1], di[o]); }
                if ((o & 3) == 3) { dr[o >> 2] = pl_level16(dr[(o >> 1) - 1], dr[o >> 1]); di[o >> 2] = pl_level16(di[(o >> 1) - 1], di[o >> 1]); }
                if ((o & 7) == 7) { dr[o >> 3] = pl_level8(dr[(o >> 2) - 1], dr[o >> 2], hi8); di[o >> 3] = pl_level8(di[(o >> 2) - 1], di[o >> 2], hi8); }
            }
        }
        const float yr = pl_level421(dr[0], dr[1], hi4), yi = pl_level421(di[0], di[1], hi4);
        const uint64_t m = (c_first - 1) * R + 1 + (uint64_t)(grp * 16 + oidx);
        if (leader && m >= ms && m < me) orow[(uint32_t)m & P.out.mask] = make_float2(yr, yi);
#pragma unroll
        for (int a = 0; a < WN - 16; ++a) acc[a] = acc[a + 16];
    }
}

// Edge scratch of one call: for every stream the ROTATED samples i0 .. i0 + len - 1 (absolute indices, i0 may be negative): zeros in
// front of the stream, the carried (already rotated) history in front of this call's buffer, the buffer's head through the exact NCO
__global__ __launch_bounds__(256) void k_pl_edge_stage(const DecimParams P_, int64_t i0, uint32_t len)
{
    const DecimParams& P = P_;
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= len) return;
    const int b = blockIdx.y;
    const int64_t i = i0 + (int64_t)j;
    float2 x = make_float2(0.f, 0.f);
    if (i >= 0) {
        const uint64_t ui = (uint64_t)i;
        if (ui >= P.n0) {
            const uint64_t r = ui - P.n0;
            if (r < P.n) {
                x = P.in[(size_t)b * P.in_stride + (size_t)r];
                const uint64_t kk = ui - P.rot_nbase;
                x = cmul_fma(x, cmul_fma(sincos_turn(P.rot_acc + ((kk >> 9) << 9) * P.rot_inc), P.rot_lo[(uint32_t)kk & 511u]));
            }
        } else {
            const uint64_t d = P.n0 - ui;
            if (d <= P.hist_len) x = P.hist[(size_t)b * P.hist_len + (P.hist_len - (uint32_t)d)];
        }
    }
    P.pl_edge[(size_t)b * P.pl_edge_stride + j] = x;
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
    const int D = P.D, E = P.pl_E, Dp = D * P.pl_R, nt = P.nt;
    // lane l owns the samples i with ((i - 1) mod D') / E = l; oldest first (tap index k descending), first term a plain product.
    // Sample e of the lane in the block whose e = 0 sample has tap k_b: tap k_b - e.  32-bit indices relative to this call's buffer.
    float vr = 0.f, vi = 0.f;
    bool first = true;
    uint32_t hi_blk = ~0u;
    float2 hi = make_float2(1.f, 0.f);
    const int64_t top = (int64_t)m * D;                          // sample index of tap 0
    const int rel_top = (int)(top - (int64_t)P.n0);              // the same, relative to in[0]
    const int64_t n_before = -(int64_t)P.n0;                     // rel < n_before: in front of the stream (zero)
    const uint64_t kk0 = P.n0 - P.rot_nbase;                     // NCO index of in[0]: coarse block blk0, fine offset lo0
    const uint64_t blk0 = kk0 >> 9; const uint32_t lo0 = (uint32_t)kk0 & 511u;
    if (E * lane < Dp) {
        int64_t k0 = (top - 1 - (int64_t)(E * lane)) % Dp;
        if (k0 < 0) k0 += Dp;
        const float2* inb = P.in ? P.in + (size_t)b * P.in_stride : nullptr;
        const float2* hb = P.hist ? P.hist + (size_t)b * P.hist_len : nullptr;
        const float2* rb = P.in_ring.p ? P.in_ring.p + (size_t)b * (P.in_ring.mask + 1u) : nullptr;
        for (int kb = (int)k0 + ((nt - 1 + E - 1 - (int)k0) / Dp) * Dp; kb >= 0; kb -= Dp) {
            for (int e = 0; e < E; ++e) {
                const int k = kb - e;
                if (k < 0 || k >= nt || E * lane + e >= Dp) continue;
                const float hk = P.pl_hraw[k];
                const int rel = rel_top - k;
                float2 x = make_float2(0.f, 0.f);
                if ((int64_t)rel >= n_before) {
                    if (inb) {
                        if (rel >= 0) {
                            x = inb[rel];
                            if (P.rot_enable) {
                                const uint32_t kk = lo0 + (uint32_t)rel;
                                if ((kk >> 9) != hi_blk) { hi_blk = kk >> 9; hi = sincos_turn(P.rot_acc + ((blk0 + hi_blk) << 9) * P.rot_inc); }
                                x = cmul_fma(x, cmul_fma(hi, P.rot_lo[kk & 511u]));
                            }
                        } else if ((uint32_t)(-rel) <= P.hist_len) {
                            x = hb[(int)P.hist_len + rel];
                        }
                    } else {
                        x = rb[(uint32_t)(P.n0 + (uint64_t)(int64_t)rel) & P.in_ring.mask];
                    }
                }
                if (first) { vr = hk * x.x; vi = hk * x.y; first = false; }
                else { vr = fmaf(hk, x.x, vr); vi = fmaf(hk, x.y, vi); }
            }
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

// geometry of the "pl" contract (oracle/orc_blocks.c orc_pl_geometry) and the instantiated kernels
struct PlGeom { int R, Dp, E, U, Upad, WU; bool ok; };
static PlGeom pl_geom(int nt, int D)
{
    PlGeom g{};
    g.R = D <= 32 ? 64 / D : 1;
    g.Dp = g.R * D;
    g.E = (g.Dp + 63) / 64;
    g.U = (nt + g.Dp - 1) / D;
    g.Upad = g.U;
    if (g.E == 1 && g.R == 1) g.ok = g.U <= 16;
    else if (g.E == 2 && g.R == 1) { g.ok = (D % 2) == 0 && g.U <= 42; g.Upad = 42; }   // (the front-end filters have 41.8 D taps: U = 42 for every D)
    else g.ok = false;
    g.WU = (g.Upad - 1) / g.R;
    return g;
}
// rule shared with oracle/orc_blocks.c orc_decim_uses_pl
bool decim_uses_pl(int nt, int D) { return pl_geom(nt, D).ok; }

size_t decim_pl_edge_len(int nt, int D)
{
    const PlGeom g = pl_geom(nt, D);
    return g.ok ? (size_t)(2 * g.WU + 4) * g.Dp : 0;   // warm-up blocks + the blocks of <= (WU + 1) R + 1 edge outputs
}

// lane tap table [Upad][E][64]: lane l, sample e of a block (r = E l + e) meets output u + 1 with h[(u + 1) D - 1 - r];
// the raw taps follow (k_decim_pl_gen reads them by index)
std::vector<float> decim_pl_layout(const std::vector<float>& h, int D)
{
    const int nt = (int)h.size();
    const PlGeom g = pl_geom(nt, D);
    std::vector<float> t((size_t)g.Upad * g.E * 64 + (size_t)nt, 0.0f);
    for (int u = 0; u < g.Upad; ++u)
        for (int e = 0; e < g.E; ++e)
            for (int l = 0; l < 64; ++l) {
                const int r = g.E * l + e;
                const int k = (u + 1) * D - 1 - r;
                if (r < g.Dp && k >= 0 && k < nt) t[((size_t)u * g.E + e) * 64 + l] = h[k];
            }
    for (int k = 0; k < nt; ++k) t[(size_t)g.Upad * g.E * 64 + k] = h[k];
    return t;
}

template <int J>
static void pl_launch_main(const DecimParams& q, uint32_t units, hipStream_t s)
{
    // LDS-DMA variant whenever the byte stream of a unit can be cut into 16-byte pieces: rows 16-byte aligned with an even sample
    // count (what qrl_demod_process demands of its callers; the edge scratch is built that way)
    const bool dma_ok = !q.pl_legacy && q.in && (reinterpret_cast<uintptr_t>(q.in) & 15u) == 0 && (q.in_stride & 1u) == 0 && (q.n & 1u) == 0 &&
                        (!q.pl_edge || ((reinterpret_cast<uintptr_t>(q.pl_edge) & 15u) == 0 && (q.pl_edge_stride & 1u) == 0));
    if (dma_ok) hipLaunchKernelGGL((k_decim_pl2<J>), dim3((units + 3) / 4), dim3(256), 0, s, q);
    else hipLaunchKernelGGL((k_decim_pl<J>), dim3((units + 3) / 4), dim3(256), 0, s, q);
}

int launch_decim_pl(const DecimParams& p, int batch, hipStream_t s)
{
    if (p.m_count == 0) return 0;
    const int D = p.D;
    const PlGeom g = pl_geom(p.nt, D);
    DecimParams q = p;
    q.pl_J = g.Upad; q.pl_E = g.E; q.pl_R = g.R;
    q.pl_hraw = p.pl_taps + (size_t)g.Upad * g.E * 64;
    const uint64_t m_end = p.m0 + p.m_count;
    uint64_t m_main = m_end, m_tail = m_end;   // [m_main, m_tail): the register kernel; the rest: one wave per output
    if (p.in && p.rot_enable) {
        // interior outputs: every sample the lanes touch lies in this call's buffer.  First block of a segment starting at output
        // ms = (cs - 1) R + 1 is cs - WU, its first sample (cs - WU - 1) D' + 1 >= n0
        const uint64_t qb = p.n0 > 1 ? (p.n0 - 1 + g.Dp - 1) / g.Dp : 0;
        m_main = (qb + g.WU) * g.R + 1;
        if (m_main < p.m0) m_main = p.m0 + (g.R - 1 - (p.m0 + g.R - 2) % g.R);   // next m == 1 (mod R)
        // last block must end inside the buffer: c D' <= n0 + n - 1
        const uint64_t c_max = (p.n0 + p.n - 1) / g.Dp;
        m_tail = c_max * g.R + 1;
        if (m_tail > m_end) m_tail = m_end;
        if (m_main > m_tail) m_main = m_tail;
    }
    bool edge_unit = false;
    if (m_main > p.m0) {
        // head edge: through the register kernel out of a staged scratch when it fits (R = 1 geometries), else one wave per output
        const int64_t c_first = (int64_t)p.m0 - g.WU;                         // (R = 1) block of output m0 minus the warm-up blocks
        const int64_t i0 = (c_first - 1) * (int64_t)g.Dp + 1;
        const int64_t i_last = (int64_t)(m_main - 1) * g.Dp;                   // last sample of the last edge block
        if (p.pl_edge && g.R == 1 && i_last - i0 + 1 <= (int64_t)p.pl_edge_cap) {
            const uint32_t len = (uint32_t)(i_last - i0 + 1);
            hipLaunchKernelGGL(k_pl_edge_stage, dim3((len + 255) / 256, batch), dim3(256), 0, s, q, i0, len);
            q.pl_edge_ms = p.m0; q.pl_edge_me = m_main;
            edge_unit = true;
        } else {
            const uint32_t cnt = (uint32_t)(m_main - p.m0);
            hipLaunchKernelGGL(k_decim_pl_gen, dim3((cnt + 3) / 4, batch), dim3(256), 0, s, q, p.m0, cnt);
        }
    }
    if (m_tail < m_end && m_tail >= m_main) {
        const uint32_t cnt = (uint32_t)(m_end - m_tail);
        hipLaunchKernelGGL(k_decim_pl_gen, dim3((cnt + 3) / 4, batch), dim3(256), 0, s, q, m_tail, cnt);
    }
    if (m_main >= m_tail && !edge_unit) return 0;
    const uint64_t total = (m_tail - m_main) * (uint64_t)batch;
    q.pl_m_begin = m_main; q.pl_m_end = m_tail;
    q.pl_batch = (uint32_t)batch;
    if (g.E == 1 && g.R == 1) {
        // segment length: a multiple of 16 blocks, long enough to keep the warm-up re-reads small, short enough to spread the
        // call over >= ~16 waves per CU, and inside the 64-entry coarse rotator table of a wave (64 x 512 samples)
        const int J = g.U;
        uint64_t S = total / (256u * 16u * 4u);
        const uint64_t s_cap = (uint64_t)((62 * 512) / D - J) / 16 * 16;
        if (S > 512) S = 512;
        if (S > s_cap) S = s_cap;
        S = S / 16 * 16;
        if (S < 16) S = 16;
        q.pl_S = (uint32_t)S;
        q.pl_nseg = (uint32_t)((m_tail - m_main + S - 1) / S);
        const uint32_t units = q.pl_nseg * (uint32_t)batch + (edge_unit ? (uint32_t)batch : 0u);
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
    // generalised geometry: two waves per SIMD (<= 256 VGPRs), ~3 segments per wave slot of the chip; a segment stays inside
    // the wave's coarse rotator table and is a multiple of 16 outputs
    uint64_t S = total / (2048u * 3u);
    const uint64_t s_cap = (uint64_t)(((PLX_NHI - 2) * 512) / g.Dp - g.WU) * g.R / 16 * 16;
    if (S > 2048) S = 2048;
    if (S > s_cap) S = s_cap;
    S = S / 16 * 16;
    if (S < 16) S = 16;
    q.pl_S = (uint32_t)S;
    q.pl_nseg = (uint32_t)((m_tail - m_main + S - 1) / S);
    const uint32_t units = q.pl_nseg * (uint32_t)batch + (edge_unit ? (uint32_t)batch : 0u);
    hipLaunchKernelGGL((k_decim_plx<2, 1, 42>), dim3((units + 3) / 4), dim3(256), 0, s, q);
    return 0;
}

}  // namespace qrl
