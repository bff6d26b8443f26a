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
#include <cstdlib>
#include <mutex>
#include "devmath.hpp"
#include "engine.hpp"

namespace qrl {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float f32x4_pm __attribute__((ext_vector_type(4)));


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

// =====================================================================================================================================
// k_decim_pm — the 1:D first stages with 32 < D <= 64 (the 1:50, 419-tap stage of gr_demod_2fsk / gmsk / 4fsk / bpsk = the C1 front end)
// as a PHASE-MAJOR product on the f32 matrix pipe.  Contract "pm" (oracle/orc_blocks.c orc_decim_fir_ccf_pm).
//
// Why: the phase-lane kernel of rounds 1-2 (lane = polyphase branch, taps in registers, transposing butterfly) issued 39 VALU
// instructions per 50-sample block -- 21 of them the tap FMAs -- and measured VALU bound (SQ_INSTS_VALU x 2 cycles = 64 % of the SIMD
// cycles at 5.1 TB/s; routing the input through LDS-DMA rings changed nothing: 6.97 - 7.06 ms against 6.86 ms).  The tap work is a matrix
// product: with the stream cut into blocks of D samples (block c = samples (c-1) D + 1 .. c D, a ROW of the input as it lies in
// memory), Z[c][j] = sum_p x~[c][p] H[p][j], H[p][j] = h[j D + D - 1 - p], and y[m] = sum_j Z[m - j][j].  One
// v_mfma_f32_16x16x4_f32 per 4 phases and component produces Z for 16 blocks x 16 block lags (J <= 16): 26 matrix instructions
// per 800 samples replace 344 VALU, and the matrix pipe runs beside the VALU (rotator) work of the other waves.
//   A operand = H^T (rows = block lag j, k = phase): 13 registers per lane, loaded once.
//   B operand = x~^T (k = phase, columns = 16 consecutive blocks): lane (q = lane >> 4, n = lane & 15) holds sample 4 s + q of block n:
//       read raw from the wave's LDS-DMA ring (ds_read_b64 at block stride 8 D bytes + 8 q: conflict free), rotated in registers
//       (exact NCO tables in LDS, as before: 14 VALU per lane and step).
//   D = Z (rows j, columns blocks): lane (q, n) holds Z[n][4 q + r], r = 0..3.  y[m] = sum_j Z[m - j][j] is a diagonal sum: DPP
//       row_shr:j inside a 16-lane row moves Z[.][j] to its output's lane, row_shl:(16 - j) collects what belongs to the NEXT group of
//       16 outputs (carried in a register), one v_permlane16_swap + one v_permlane32_swap fold the four lane rows (both components
//       at once): 44 VALU per 16 outputs instead of 70.
// Data path = the LDS-DMA ring of round 3's experiments (tools/ubench/stream_lds.hip: 7.07 TB/s for wave-private rings, non-temporal
// 1 KiB pieces): a wave streams its segment through a private ring of 8 KiB; a group of 16 blocks (6400 bytes at D = 50) is copied
// to registers as soon as it has landed, which frees its slots for the next pieces -- the ring is only the landing zone, ~7 KiB per
// wave stay in flight.  No barrier in the loop; counted s_waitcnt vmcnt orders the DMA against the wave's own reads.
// Groups sit on an ABSOLUTE grid (blocks 16 G .. 16 G + 15), so the value of output m depends on m alone (chunk invariance).
// Diagonal sums of one matrix result (lane (q, n) holds Z[n][4 q + r], r = 0..3), for all four lane rows at once -- 15 DPP moves /
// adds and 2 plain moves per component instead of one masked shift-add per lag and row:
//   1. the four lags of a row differ by 0..3 lanes whatever the row:  L[n] = ((z0[n] + z1[n-1]) + z2[n-2]) + z3[n-3]  (row_shr, +0 from
//      outside the row), and what the shifts push out of the group  H[k] = (z1[15+k] + z2[14+k]) + z3[13+k], k = 0..2 (row_shl);
//   2. row q then only has to move by 4 q lanes: R = row_shr:4q of L (in-row terms), C = row_shl:(16 - 4q) of L in the lanes below
//      4 q joined with row_shr:4q of H above them (the carries into the next group): moves under a row mask, no arithmetic.
// One asm statement: the compiler keeps v_mov_b32_dpp + v_add_f32 apart, and its hazard recogniser does not look inside asm -- the
// leading s_nop 1 and the instruction order respect the gfx9 rule "VALU writes a VGPR, DPP reads it: 2 wait states" (H is read three
// instructions after its last write, L four).
__device__ __forceinline__ void pm_diag(const f32x4_pm& z, float& R, float& C)
{
    float L, H;
    asm("s_nop 1\n\t"
        "v_mov_b32_dpp %[H], %[z1] row_shl:15 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_add_f32_dpp %[H], %[z2], %[H] row_shl:14 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_add_f32_dpp %[H], %[z3], %[H] row_shl:13 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_add_f32_dpp %[L], %[z1], %[z0] row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_add_f32_dpp %[L], %[z2], %[L] row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_add_f32_dpp %[L], %[z3], %[L] row_shr:3 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_mov_b32 %[C], %[H]\n\t"
        "v_mov_b32 %[R], %[L]\n\t"
        "v_mov_b32_dpp %[C], %[H] row_shr:4 row_mask:0x2 bank_mask:0xf bound_ctrl:0\n\t"
        "v_mov_b32_dpp %[C], %[H] row_shr:8 row_mask:0x4 bank_mask:0xf bound_ctrl:0\n\t"
        "v_mov_b32_dpp %[C], %[H] row_shr:12 row_mask:0x8 bank_mask:0xf bound_ctrl:0\n\t"
        "v_mov_b32_dpp %[C], %[L] row_shl:12 row_mask:0x2 bank_mask:0xf\n\t"
        "v_mov_b32_dpp %[C], %[L] row_shl:8 row_mask:0x4 bank_mask:0xf\n\t"
        "v_mov_b32_dpp %[C], %[L] row_shl:4 row_mask:0x8 bank_mask:0xf\n\t"
        "v_mov_b32_dpp %[R], %[L] row_shr:4 row_mask:0x2 bank_mask:0xf bound_ctrl:0\n\t"
        "v_mov_b32_dpp %[R], %[L] row_shr:8 row_mask:0x4 bank_mask:0xf bound_ctrl:0\n\t"
        "v_mov_b32_dpp %[R], %[L] row_shr:12 row_mask:0x8 bank_mask:0xf bound_ctrl:0"
        : [R] "=&v"(R), [C] "=&v"(C), [L] "=&v"(L), [H] "=&v"(H)
        : [z0] "v"(z[0]), [z1] "v"(z[1]), [z2] "v"(z[2]), [z3] "v"(z[3]));
}
__device__ __forceinline__ void pm_glds16(const void* gsrc, uint32_t lds_dst)
{
    // one LDS-DMA piece, non-temporal: 64 lanes x 16 B from per-lane global addresses to LDS[lds_dst + 16 lane].  M0 (compiler
    // reserved) is saved and restored inside the statement (cdna_hip_programming.md, "LDS-DMA recipe")
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
typedef const __attribute__((address_space(3))) v2f* pm_lds_f2;
__device__ __forceinline__ float2 pm_lds(uint32_t addr)   // ds_read_b64 from a raw LDS byte address
{
    const v2f v = *(pm_lds_f2)(uintptr_t)addr;
    return make_float2(v.x, v.y);
}
__device__ __forceinline__ uint32_t pm_lds_addr(const void* p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p; }
template <int N> __device__ __forceinline__ void pm_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ void pm_wait_vm_dyn(uint32_t allowed)   // wave uniform; at most `allowed` DMA pieces may still be in flight
{
    switch (allowed) {
    case 0: pm_wait_vm<0>(); break;   case 1: pm_wait_vm<1>(); break;   case 2: pm_wait_vm<2>(); break;   case 3: pm_wait_vm<3>(); break;
    case 4: pm_wait_vm<4>(); break;   case 5: pm_wait_vm<5>(); break;   case 6: pm_wait_vm<6>(); break;   case 7: pm_wait_vm<7>(); break;
    case 8: pm_wait_vm<8>(); break;   case 9: pm_wait_vm<9>(); break;   case 10: pm_wait_vm<10>(); break; case 11: pm_wait_vm<11>(); break;
    case 12: pm_wait_vm<12>(); break; case 13: pm_wait_vm<13>(); break; case 14: pm_wait_vm<14>(); break; default: pm_wait_vm<15>(); break;
    }
}

#ifndef QRL_PM_ABL
#define QRL_PM_ABL 0   // developer builds only (tools/pl_variants.sh): bit 0 no output store, bit 1 no rotator, bit 2 no MFMA, bit 3 no diagonal sums (wrong results; timing ablations)
#endif
#ifdef QRL_PM_PROF
// developer build (tools/pl_variants.sh -DQRL_PM_PROF): shader-clock ticks per phase of the group loop, summed over every wave
__device__ unsigned long long g_pm_prof[8];
#define PM_STAMP(k) do { const unsigned long long tn_ = __builtin_readcyclecounter(); pc[k] += tn_ - tprev; tprev = tn_; } while (0)
#else
#define PM_STAMP(k) do { } while (0)
#endif
// J > 16 (the device-rate front ends: 41.8 D taps = 42 block lags): the lags come in NT = 3 TILES of 16, j = 16 t + j'.  Tile t of
// block group G and tile 0 of group G + t feed the SAME outputs (block 16 G + n, lag 16 t + j' -> output 16 (G + t) + n + j'), so the
// accumulator of an output group is handed from tile to tile through the C operand: it starts with tile 2 of group G - 2, takes tile 1
// of group G - 1 and is finished by tile 0 of group G -- one diagonal sum per group and component, whatever the number of tiles.
// 42 matrix instructions per 16 blocks of D samples = 0.094 per sample whatever D: the kernel is bound by the matrix pipe at about
// 4.1 - 4.4 TB/s of input (tools/ubench/stream_lds.hip W4), against 2.5 - 2.7 TB/s for the banded-Toeplitz form of rounds 1-2 (26 % of
// its matrix work multiplied zero taps, every tile went through LDS twice) and for the VALU phase-lane kernel at 100:1.
constexpr int PM_NHI = 64;   // coarse rotator entries per wave (64 x 512 samples): the wave re-bases and refills its own table as it walks
// K1 (round 6): D = 4 (NS - 1) + 1 -- the block's last step holds ONE phase (D = 25: 25 phases in 7 steps of 4).  The step then runs as
// v_mfma_f32_4x4x1_16B_f32 (K = 1, 8 cycles) instead of v_mfma_f32_16x16x4_f32 (K = 4, 32 cycles, three of its four products 0 x 0):
// 36 x 32 + 6 x 8 = 1200 instead of 42 x 32 = 1344 matrix-pipe cycles per group at D = 25.  Sixteen 4 x 4 outer products per instruction:
// batch (lanes 4 b .. 4 b + 3), A[b][i] x B[b][j] -> register i of lane 4 b + j.  With batch b = 4 q + (n >> 2) that IS the accumulator
// layout of the 16x16x4 instructions (lane (q, n) register r = Z[block n][lag 4 q + r]): A = tap H[D - 1][16 t + 4 q + (n & 3)],
// B = the rotated last sample of block n in EVERY lane row (row 0 broadcast with two lane-row swaps).  One product added with one
// rounding: the same link of the pm chain (the three links it drops added 0 x 0).
template <int J, int NS, int RP, int NW, bool K1 = false>
__global__ __launch_bounds__(NW * 64)
void k_decim_pm(const DecimParams P_)
{
    constexpr int NHI = PM_NHI;
    constexpr int NT = (J + 15) / 16;
    static_assert(NT == 1 || NT == 3, "one lag tile (J <= 16) or three (32 < J <= 48)");
    const DecimParams& P = P_;
    // dynamic LDS (the kernel has no static allocation, so it starts at LDS byte 0): rings first -- ring of wave w at byte w * ring
    // size, which the address arithmetic below relies on --, then the tables
    extern __shared__ __align__(16) unsigned char pm_smem[];
    unsigned char* ring_all = pm_smem;
    float2* t_lo = reinterpret_cast<float2*>(pm_smem + NW * RP * 1024);            // fine rotator table; entry 0 is exactly (1, 0): what edge units read through index mask 0
    float2 (*t_hi_all)[NHI] = reinterpret_cast<float2 (*)[NHI]>(t_lo + 512);      // coarse rotator table of each wave's segment (NHI x 512 samples)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int k = tid; k < 512; k += NW * 64) t_lo[k] = P.rot_lo[k];

    // unit = (stream, segment); behind the regular units one EDGE unit per stream (outputs pl_edge_ms .. pl_edge_me out of the staged,
    // already rotated scratch: identity phasors)
    const uint32_t unit = blockIdx.x * (uint32_t)NW + (uint32_t)wave;
    const uint32_t B = P.pl_batch;
    const uint32_t nreg = P.pl_nseg * B;
    const bool edge = unit >= nreg;
    const uint32_t nsg = P.pl_nseg ? P.pl_nseg : 1u;
    const uint32_t b = edge ? unit - nreg : unit / nsg, seg = edge ? 0u : unit - b * nsg;
    const bool active = edge ? (b < B && P.pl_edge_me > P.pl_edge_ms) : true;
    const int D = P.D;
    const uint64_t ms = edge ? P.pl_edge_ms : P.pl_m_begin + (uint64_t)seg * P.pl_S;
    const uint64_t me = edge ? P.pl_edge_me : (ms + P.pl_S < P.pl_m_end ? ms + P.pl_S : P.pl_m_end);
    // absolute 16-block groups: the first one holds block ms - (J - 1), the oldest block output ms needs
    const int64_t cfs = (int64_t)ms - (J - 1);
    const int64_t G0 = cfs >= 0 ? cfs / 16 : -((-cfs + 15) / 16);
    const int ngrp = active ? (int)((int64_t)((me - 1) / 16) - G0) + 1 : 0;
    const int64_t i_first_s = (G0 * 16 - 1) * (int64_t)D + 1;        // first sample of block 16 G0 (regular units: >= n0; edge units: the scratch starts here)
    const uint64_t i_first = (uint64_t)i_first_s;
    uint32_t kb0 = edge ? 0u : (uint32_t)((i_first - P.rot_nbase) >> 9);          // 512-sample block of the coarse table's entry 0
    if (active) t_hi_all[wave][lane] = edge ? make_float2(1.f, 0.f) : sincos_turn(P.rot_acc + ((uint64_t)(kb0 + (uint32_t)lane) << 9) * P.rot_inc);
    __syncthreads();
    if (!active) return;

    float a[NT][NS];                                                 // A operands: H[p = 4 s + (lane >> 4)][j = 16 t + (lane & 15)]
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int s = 0; s < NS; ++s) a[t][s] = P.pl_taps[(t * NS + s) * 64 + lane];
    const int q = lane >> 4, nn = lane & 15;
    float a1[NT];                                                    // K1: taps of the block's last phase in the 4x4x1 layout (see above)
#pragma unroll
    for (int t = 0; t < NT; ++t) a1[t] = K1 ? __shfl(a[t][NS - 1], 4 * q + (nn & 3), 64) : 0.f;   // held by lane (row 0, column 4 q + (n & 3)) of the K = 4 layout
    // the wave's byte stream: row = the stream's buffer (or its edge scratch), first byte of block 16 G0 at row + off0
    const unsigned char* rowp = reinterpret_cast<const unsigned char*>(edge ? P.pl_edge + (size_t)b * P.pl_edge_stride : P.in + (size_t)b * P.in_stride);
    const uint64_t off0 = edge ? 0ull : (uint64_t)(i_first - P.n0) * 8ull;
    const uint64_t row_bytes = edge ? (uint64_t)P.pl_edge_stride * 8ull : (uint64_t)P.n * 8ull;   // multiples of 16 (even sample counts)
    const uint32_t o0 = (uint32_t)(off0 & 127u);                     // offset of the first block inside piece 0
    const uint64_t a_off = off0 - o0;                                // piece 0 starts here (128-byte aligned relative to the row)
    const uint32_t GB = 16u * (uint32_t)D * 8u;                      // bytes per group
    const uint32_t npieces = (o0 + (uint32_t)ngrp * GB + 1023u) >> 10;
    const uint32_t q_safe = (uint32_t)((row_bytes - a_off) >> 10);   // pieces [0, q_safe) lie inside the row; later ones are clamped to its last 16 bytes (never consumed)
    const uint32_t rbase = pm_lds_addr(ring_all) + (uint32_t)wave * (RP * 1024u);   // multiple of the ring size
    const uint32_t tlo_base = pm_lds_addr(t_lo);
    const uint32_t thi_base = pm_lds_addr(t_hi_all) + (uint32_t)wave * (NHI * 8u);
    const unsigned char* gp = rowp + a_off + (size_t)lane * 16;      // this lane's 16 bytes of the next piece
    const unsigned char* last16 = rowp + row_bytes - 16;
    uint32_t issued = 0;
    auto issue_upto = [&](uint32_t want) {                           // wave uniform
        while (issued < want) {
            const uint32_t dst = rbase + (issued & (RP - 1)) * 1024u;
            if (issued < q_safe) pm_glds16(gp, dst);
            else pm_glds16(gp < last16 ? gp : last16, dst);
            gp += 1024;
            ++issued;
        }
    };
    const uint32_t k0 = edge ? 0u : (uint32_t)(i_first - P.rot_nbase) - (kb0 << 9);   // < 512
    const uint32_t tl_mask = edge ? 0u : 4095u;
    const uint32_t lane_byte = ((uint32_t)nn * (uint32_t)D + (uint32_t)q) * 8u;      // this lane's sample of step 0 inside a group
    const bool last_valid = 4 * (NS - 1) + q < D;                    // step NS - 1 reaches past the block for the upper lane rows
    float2* orow = P.out.p + ((size_t)b * (P.out_row_mul_m1 + 1u) + P.out_row_add) * (P.out.mask + 1u);
    float Cpr = 0.f, Cpi = 0.f;                                      // carries of the previous group, per lane row
    f32x4_pm nr1 = {0.f, 0.f, 0.f, 0.f}, ni1 = nr1, nr2 = nr1, ni2 = nr1;   // (three tiles) the accumulators of the next two output groups

#ifdef QRL_PM_PROF
    unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_readcyclecounter();
#endif
    issue_upto(npieces < RP ? npieces : RP);                         // (piece 0 is the first piece of group 0)
    uint32_t gbytes = o0;                                            // byte offset of the current group in the wave's stream
    uint32_t kb8 = (k0 * 8u) + lane_byte;                            // 8 x NCO index (relative to the coarse table) of this lane's step-0 sample
    uint32_t kg8 = k0 * 8u;                                          // the same for the group's first sample (wave uniform)
    for (int g = 0; g < ngrp; ++g) {
        if (((kg8 + GB) >> 12) >= (uint32_t)(NHI - 1)) {             // the group would leave the coarse table: re-base it (wave private, no barrier)
            const uint32_t sh = kg8 >> 12;
            kb0 += sh; kg8 -= sh << 12; kb8 -= sh << 12;
            t_hi_all[wave][lane] = edge ? make_float2(1.f, 0.f) : sincos_turn(P.rot_acc + ((uint64_t)(kb0 + (uint32_t)lane) << 9) * P.rot_inc);
        }
        {   // everything up to the end of this group must have landed; younger pieces may stay in flight
            uint32_t need = (gbytes + GB + 1023u) >> 10;
            need = need < npieces ? need : npieces;
            pm_wait_vm_dyn(issued - need);
        }
        PM_STAMP(0);
        float2 x[NS];
        {
            const uint32_t xb = gbytes + lane_byte;
#pragma unroll
            for (int s = 0; s < NS; ++s) x[s] = pm_lds(rbase | ((xb + 32u * (uint32_t)s) & (RP * 1024u - 1u)));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the group is in registers: its ring slots are free
        PM_STAMP(1);
        {
            const uint32_t first_next = (gbytes + GB) >> 10;         // first piece the next group still needs
            const uint32_t cap = first_next + RP;
            issue_upto(cap < npieces ? cap : npieces);
        }
        PM_STAMP(2);
        f32x4_pm zr[NT], zi[NT];                                     // [t]: gets tile t now; [0] is complete after this group
        zr[NT - 1] = f32x4_pm{0.f, 0.f, 0.f, 0.f}; zi[NT - 1] = zr[NT - 1];
        if constexpr (NT == 3) { zr[0] = nr1; zi[0] = ni1; zr[1] = nr2; zi[1] = ni2; }
        // rotator: phasor of sample k = T_hi[k >> 9] (x) T_lo[k & 511], byte addressed; the table reads run two steps ahead, so the rotation never
        // waits for the LDS.  (Pinning the rotation of step s + 1 between the matrix instructions of step s with sched_group_barrier, and distinct
        // s_setprio per resident workgroup, were measured without effect -- docs/KERNELS.md 1 -- and deleted in round 6.)
        struct Ph { float2 lo, hi; };
        auto tab = [&](int s_) -> Ph {
            const uint32_t kk = kb8 + 32u * (uint32_t)s_;
            return Ph{pm_lds(tlo_base + (kk & tl_mask)), pm_lds(thi_base + ((kk >> 12) << 3))};
        };
        auto rot = [&](int s_, const Ph& ph) -> float2 {
#if QRL_PM_ABL & 2
            float2 xs_ = x[s_]; (void)ph;
#else
            float2 xs_ = cmul_fma(x[s_], cmul_fma(ph.hi, ph.lo));
#endif
            if (s_ == NS - 1 && !last_valid) xs_ = make_float2(0.f, 0.f);   // phases >= D: zero taps AND zero samples
            return xs_;
        };
        float2 xs = rot(0, tab(0));
        Ph pn = tab(NS > 1 ? 1 : 0);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            float2 xn = xs;
            Ph pnn = pn;
            // (scheduling barriers pin the order: the reads for the step after next are issued HERE, a whole step before their use --
            // otherwise the scheduler sinks every read down to its use and the rotation waits for the LDS)
            if (s + 2 < NS) pnn = tab(s + 2);
            if (s + 1 < NS) xn = rot(s + 1, pn);
            pn = pnn;
            if (K1 && s == NS - 1) {
                // the block's last sample, row 0 of the rotated operand, into every lane row: (r0, r1, r2, r3) -> (r0, r1, r0, r1) -> (r0, r0, r0, r0)
                const auto h32r = __builtin_amdgcn_permlane32_swap(__float_as_uint(xs.x), __float_as_uint(xs.x), false, false);
                const auto h32i = __builtin_amdgcn_permlane32_swap(__float_as_uint(xs.y), __float_as_uint(xs.y), false, false);
                const auto h16r = __builtin_amdgcn_permlane16_swap(h32r[0], h32r[0], false, false);
                const auto h16i = __builtin_amdgcn_permlane16_swap(h32i[0], h32i[0], false, false);
                const float bx = __uint_as_float(h16r[0]), by = __uint_as_float(h16i[0]);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    zr[t] = __builtin_amdgcn_mfma_f32_4x4x1f32(a1[t], bx, zr[t], 0, 0, 0);
                    zi[t] = __builtin_amdgcn_mfma_f32_4x4x1f32(a1[t], by, zi[t], 0, 0, 0);
                }
            } else
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#if QRL_PM_ABL & 4
                zr[t][s & 3] = fmaf(a[t][s], xs.x, zr[t][s & 3]); zi[t][s & 3] = fmaf(a[t][s], xs.y, zi[t][s & 3]);
#else
                zr[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][s], xs.x, zr[t], 0, 0, 0);
                zi[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][s], xs.y, zi[t], 0, 0, 0);
#endif
            }
            xs = xn;
        }
        PM_STAMP(3);
        if constexpr (NT == 3) { nr1 = zr[1]; ni1 = zi[1]; nr2 = zr[2]; ni2 = zi[2]; }
        // y[16 G + n'] = sum_j' Z[n' - j'][j']: per lane row the in-row terms (R) and the carries into the next group (C)
        float Rr, Ri, Cr, Ci;
#if QRL_PM_ABL & 8
        Rr = zr[0][0] + zr[0][1] + zr[0][2] + zr[0][3]; Ri = zi[0][0] + zi[0][1] + zi[0][2] + zi[0][3]; Cr = Rr; Ci = Ri;
#else
        pm_diag(zr[0], Rr, Cr);
        pm_diag(zi[0], Ri, Ci);
#endif
        const float Vr = Rr + Cpr, Vi = Ri + Cpi;
        Cpr = Cr; Cpi = Ci;
        // fold the four lane rows: (V_0 + V_1) + (V_2 + V_3); rows 0 / 1 of the result = real / imaginary part of the 16 outputs
        const auto w16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(Vr), __float_as_uint(Vi), false, false);
        const float w = __uint_as_float(w16[0]) + __uint_as_float(w16[1]);
        const auto w32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(w), __float_as_uint(w), false, false);
        const float y = __uint_as_float(w32[0]) + __uint_as_float(w32[1]);
        const int64_t m = (G0 + g) * 16 + nn;
#if QRL_PM_ABL & 1
        if (y == 12345.678f) reinterpret_cast<float*>(orow)[0] = y;
#else
        if (lane < 32 && m >= (int64_t)ms && m < (int64_t)me)
            reinterpret_cast<float*>(orow + ((uint32_t)m & P.out.mask))[q] = y;
#endif
        gbytes += GB;
        kb8 += GB; kg8 += GB;
        PM_STAMP(4);
    }
#ifdef QRL_PM_PROF
    if (lane == 0) { for (int k = 0; k < 5; ++k) atomicAdd(&g_pm_prof[k], pc[k]); atomicAdd(&g_pm_prof[7], (unsigned long long)ngrp); }
#endif
    pm_wait_vm<0>();   // no DMA may still be writing this wave's ring when the workgroup's LDS is handed on
}

// Contract "pm", one THREAD per output with checked fetches: zero in front of the stream, carried (already rotated) history in front of
// this call's buffer, the caller's buffer with the exact NCO, or an engine ring.  Call edges that do not fit the staged scratch, the
// outputs behind the last whole block of a call, and the second-stage form (input = an engine ring).
__global__ __launch_bounds__(256) void k_decim_pm_gen(const DecimParams P_, uint64_t m_first, uint32_t count)
{
    const DecimParams& P = P_;
    const uint32_t o = blockIdx.x * 256u + threadIdx.x;
    if (o >= count) return;
    const int b = blockIdx.y;
    const uint64_t m = m_first + o;
    const int D = P.D, nt = P.nt, J = (nt + D - 1) / D;
    const float2* inb = P.in ? P.in + (size_t)b * P.in_stride : nullptr;
    const float2* hb = P.hist ? P.hist + (size_t)b * P.hist_len : nullptr;
    const float2* rb = P.in_ring.p ? P.in_ring.p + (size_t)b * (P.in_ring.mask + 1u) : nullptr;
    const uint64_t kk0 = P.n0 - P.rot_nbase;                     // NCO index of in[0]
    uint64_t hi_blk = ~0ull;
    float2 hi = make_float2(1.f, 0.f);
    const int NT = (J + 15) / 16;
    // Z(c, j') of the contract: one chain over the tiles t = NT-1 .. 0 and the D samples of block c - 16 t, p ascending
    auto zval = [&](int64_t c, int jp, float& zr_, float& zi_) {
        float zr = 0.f, zi = 0.f;
        for (int t = NT - 1; t >= 0; --t) {
            const int j = 16 * t + jp;
            const int64_t cb = c - 16 * (int64_t)t;
            for (int p = 0; p < D; ++p) {
                const int64_t k = (int64_t)j * D + D - 1 - p;
                const int64_t i = (cb - 1) * (int64_t)D + 1 + p;
                const float h = k < nt ? P.pl_hraw[k] : 0.f;
                float2 x = make_float2(0.f, 0.f);
                if (i >= 0) {
                    const uint64_t ui = (uint64_t)i;
                    if (inb) {
                        if (ui >= P.n0) {
                            const uint64_t rel = ui - P.n0;
                            if (rel < P.n) {
                                x = inb[rel];
                                if (P.rot_enable) {
                                    const uint64_t kk = kk0 + rel;
                                    if ((kk >> 9) != hi_blk) { hi_blk = kk >> 9; hi = sincos_turn(P.rot_acc + (hi_blk << 9) * P.rot_inc); }
                                    x = cmul_fma(x, cmul_fma(hi, P.rot_lo[(uint32_t)kk & 511u]));
                                }
                            }
                        } else {
                            const uint64_t d = P.n0 - ui;
                            if (d <= P.hist_len) x = hb[P.hist_len - (uint32_t)d];
                        }
                    } else if (ui < P.n0 + P.n) {
                        x = rb[(uint32_t)ui & P.in_ring.mask];
                    }
                }
                zr = fmaf(h, x.x, zr); zi = fmaf(h, x.y, zi);
            }
        }
        zr_ = zr; zi_ = zi;
    };
    const int64_t G = (int64_t)(m >> 4);
    const int np = (int)(m & 15u);
    float Vr[4], Vi[4];
    for (int qq = 0; qq < 4; ++qq) {
        float Rr = 0.f, Ri = 0.f, Cr = 0.f, Ci = 0.f;
        // L_q(Gx)[n] = ((Z(n, 4q) + Z(n-1, 4q+1)) + Z(n-2, 4q+2)) + Z(n-3, 4q+3), blocks outside the group = +0
        auto Lsum = [&](int64_t Gx, int n, float& lr, float& li) {
            for (int r = 0; r < 4; ++r) {
                float zr = 0.f, zi = 0.f;
                if (n - r >= 0) zval(16 * Gx + n - r, 4 * qq + r, zr, zi);
                if (r == 0) { lr = zr; li = zi; } else { lr = lr + zr; li = li + zi; }
            }
        };
        if (np >= 4 * qq) Lsum(G, np - 4 * qq, Rr, Ri);
        if (np < 4 * qq) Lsum(G - 1, np + 16 - 4 * qq, Cr, Ci);
        else if (np - 4 * qq <= 2) {                              // H_q(G - 1)[k] = (Z(15+k, 4q+1) + Z(14+k, 4q+2)) + Z(13+k, 4q+3)
            const int kq = np - 4 * qq;
            for (int r = 1; r < 4; ++r) {
                float zr = 0.f, zi = 0.f;
                if (kq <= r - 1) zval(16 * (G - 1) + 16 - r + kq, 4 * qq + r, zr, zi);
                if (r == 1) { Cr = zr; Ci = zi; } else { Cr = Cr + zr; Ci = Ci + zi; }
            }
        }
        Vr[qq] = Rr + Cr; Vi[qq] = Ri + Ci;
    }
    P.out.p[((size_t)b * (P.out_row_mul_m1 + 1u) + P.out_row_add) * (P.out.mask + 1u) + ((uint32_t)m & P.out.mask)] =
        make_float2((Vr[0] + Vr[1]) + (Vr[2] + Vr[3]), (Vi[0] + Vi[1]) + (Vi[2] + Vi[3]));
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
    if (g.E == 1 && g.R == 1) g.ok = false;   // 32 < D <= 64: the phase-major matrix-pipe kernel (decim_uses_pm) or the generic paths
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
            hipLaunchKernelGGL(k_pl_edge_stage, dim3((len + 255) / 256, batch), dim3(256), 0, p.pre_stream ? p.pre_stream : s, q, i0, len);
            if (p.pre_stream) { (void)hipEventRecord(p.pre_event, p.pre_stream); (void)hipStreamWaitEvent(s, p.pre_event, 0); }
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

#ifdef QRL_PM_PROF
extern "C" void qrl_pm_prof_read(unsigned long long* out8)
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_pm_prof), 8 * sizeof(unsigned long long));
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pm_prof), z, sizeof z);
}
#endif
// ---- "pm" contract: rule (shared with oracle/orc_blocks.c orc_decim_uses_pm), tables, launcher ---------------------------------------
bool decim_uses_pm(int nt, int D)
{
    const int J = (nt + D - 1) / D, NS = (D + 3) / 4;
    if (D > 32 && D <= 52 && J <= 16) return true;                // one lag tile: the 1:50 first stages of a 1 Msps device
    // up to three lag tiles: the device-rate front ends (41.8 D taps) at 10:1, 20:1, 25:1, 50:1, 100:1
    return J > 16 && J <= 48 && (NS == 3 || NS == 5 || NS == 7 || NS == 13 || NS == 25);
}
// block lags the instantiated kernel walks (its table is zero beyond the filter's own J): segment alignment, edge unit and look-back
// are all counted in THESE lags
static int pm_kernel_lags(int nt, int D)
{
    const int J = (nt + D - 1) / D, NS = (D + 3) / 4;
    if (J <= 16) return J == 9 && NS == 13 ? 9 : 16;
    return J <= 42 ? 42 : 48;
}
size_t decim_pm_edge_len(int nt, int D)
{
    // the edge unit of a call starts at the 16-block group of block m0 - (J - 1) and ends with the last output in front of the first
    // aligned segment: at most 15 + (J - 1) + (J + 16) blocks
    const int J = pm_kernel_lags(nt, D);
    return decim_uses_pm(nt, D) ? (size_t)(((2 * J + 32) * D + 1) & ~1) : 0;
}
uint32_t decim_pm_lookback(int nt, int D) { return (uint32_t)((pm_kernel_lags(nt, D) + 18) * D); }
// A-operand table [NT][NS][64]: lane l (j = 16 t + (l & 15), k = l >> 4) of step s, tile t holds H[p = 4 s + k][j] = h[j D + D - 1 - p];
// the raw taps follow
static int pm_tiles(int J) { return J <= 16 ? 1 : 3; }
std::vector<float> decim_pm_layout(const std::vector<float>& h, int D)
{
    const int nt = (int)h.size(), J = (nt + D - 1) / D, NS = (D + 3) / 4, NT = pm_tiles(J);
    std::vector<float> t((size_t)NT * NS * 64 + (size_t)nt, 0.0f);
    for (int tt = 0; tt < NT; ++tt)
        for (int s = 0; s < NS; ++s)
            for (int l = 0; l < 64; ++l) {
                const int j = 16 * tt + (l & 15), p = 4 * s + (l >> 4), k = j * D + D - 1 - p;
                if (j < J && p < D && k < nt) t[((size_t)tt * NS + s) * 64 + l] = h[k];
            }
    for (int k = 0; k < nt; ++k) t[(size_t)NT * NS * 64 + k] = h[k];
    return t;
}
#ifndef QRL_PM_RP
#define QRL_PM_RP 8
#endif
#ifndef QRL_PM_NW
#define QRL_PM_NW 4
#endif
#ifndef QRL_PM_RP_SMALL
#define QRL_PM_RP_SMALL 8   // ring pieces per wave for D <= 28 (a group is <= 3.5 KiB there)
#endif
// Segments: a multiple of 16 outputs each (every segment then starts on a group boundary).  Every segment re-reads `warm` blocks of
// warm-up, and the call ends when the last wave does: with `cap` waves resident on the chip the cost of a split into nseg segments per
// stream is  ceil(nseg x batch / cap)  rounds of  S + warm + 16  blocks.  Take the cheapest split; ties go to the finer one.
static uint32_t pm_pick_segment(uint64_t M, uint32_t batch, uint32_t cap, uint32_t warm)
{
    if (M <= 16) return 16;
    uint64_t best_cost = ~0ull; uint32_t best_S = 16;
    uint32_t n_lo = (uint32_t)((M + 4095) / 4096);
    const uint32_t n_hi = (uint32_t)((M + 127) / 128);
    for (uint32_t nseg = n_lo; nseg <= n_hi; ++nseg) {
        const uint32_t S = (uint32_t)(((M + nseg - 1) / nseg + 15) / 16 * 16);
        const uint64_t units = ((M + S - 1) / S) * (uint64_t)batch;
        const uint64_t cost = ((units + cap - 1) / cap) * (uint64_t)(S + warm + 16);
        if (cost <= best_cost) { best_cost = cost; best_S = S; }
    }
    return best_S;
}
template <int J, int NS, int RP = QRL_PM_RP, bool K1 = false>
static int pm_launch_main(DecimParams& q, uint64_t M, uint32_t batch, bool edge_unit, hipStream_t s)
{
    // RP = ring pieces per wave (a power of two: one 16-block group -- 128 D bytes -- + what is in flight).
    // 4 waves per workgroup; at D = 50 four workgroups per CU (156 KB of LDS).  8-wave workgroups (two per CU, 144 KB: room for slim
    // recursion kernels of the previous call beside them) were measured: 7.24 ms against 6.57 ms alone, and no gain in the overlapped mode.
    constexpr int NW = QRL_PM_NW;
    const auto kern = k_decim_pm<J, NS, RP, NW, K1>;
    size_t lds = (size_t)NW * RP * 1024 + 512 * sizeof(float2) + NW * PM_NHI * sizeof(float2);
    // experiment (VERDICT r5 #4a): pad the dynamic LDS so that fewer front-end workgroups share a CU and a tail-chain workgroup of the previous call always
    // finds room beside them (QRL_PM_LDS_KB = total KB per workgroup; 45 -> three per CU with 25 KB to spare)
    static const size_t lds_min = [] { const char* e = std::getenv("QRL_PM_LDS_KB"); return e ? (size_t)std::atoi(e) * 1024 : (size_t)0; }();
    if (lds_min > lds && lds_min <= 160 * 1024) lds = lds_min;
    // occupancy and CU count per (instantiation, device): a second device in the process gets its own LDS attribute (dyn_lds_limit
    // de-duplicates per kernel and device) and its own geometry; first calls may race, hence the lock
    static std::mutex mu; static int wg_cache[16], cu_cache[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return -1;
    if (dyn_lds_limit(reinterpret_cast<const void*>(kern), 160 * 1024) != hipSuccess) return -1;
    int wg_per_cu, n_cu;
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!wg_cache[dev]) {
            int nb = 0; hipDeviceProp_t pr;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, NW * 64, lds) != hipSuccess || nb < 1) nb = 1;
            if (hipGetDeviceProperties(&pr, dev) != hipSuccess) return -1;
            cu_cache[dev] = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
            wg_cache[dev] = nb;
        }
        wg_per_cu = wg_cache[dev]; n_cu = cu_cache[dev];
    }
    const uint32_t S = pm_pick_segment(M, batch, (uint32_t)(wg_per_cu * n_cu * NW), (uint32_t)(J - 1));
    q.pl_S = S;
    q.pl_nseg = M ? (uint32_t)((M + S - 1) / S) : 0u;
    const uint32_t units = q.pl_nseg * batch + (edge_unit ? batch : 0u);
    hipLaunchKernelGGL(kern, dim3((units + NW - 1) / NW), dim3(NW * 64), lds, s, q);
    return 0;
}
int launch_decim_pm(const DecimParams& p, int batch, hipStream_t s)
{
    if (p.m_count == 0) return 0;
    const int D = p.D, nt = p.nt, NS = (D + 3) / 4;
    const int J = pm_kernel_lags(nt, D);
    DecimParams q = p;
    q.pl_J = J; q.pl_E = 1; q.pl_R = 1;
    q.pl_hraw = p.pl_taps + (size_t)pm_tiles(J) * NS * 64;
    const uint64_t m_end = p.m0 + p.m_count;
    auto gen = [&](uint64_t first, uint64_t last) {
        if (last > first) hipLaunchKernelGGL(k_decim_pm_gen, dim3((uint32_t)((last - first + 255) / 256), batch), dim3(256), 0, s, q, first, (uint32_t)(last - first));
    };
    // the matrix kernel streams rows of the caller's buffer as 16-byte LDS-DMA pieces: rows 16-byte aligned, even sample counts
    // (what qrl_demod_process demands; the edge scratch is built that way).  Anything else: one thread per output.
    const bool dma_ok = p.in && p.rot_enable && (reinterpret_cast<uintptr_t>(p.in) & 15u) == 0 && (p.in_stride & 1u) == 0 && (p.n & 1u) == 0 &&
                        p.pl_edge && (reinterpret_cast<uintptr_t>(p.pl_edge) & 15u) == 0 && (p.pl_edge_stride & 1u) == 0;
    if (!dma_ok) { gen(p.m0, m_end); return 0; }
    // interior segments start at outputs m == J - 1 (mod 16): their oldest block opens a 16-block group, and it lies in the buffer
    const uint64_t qb = p.n0 > 1 ? (p.n0 - 1 + D - 1) / D : 0;             // block qb + 1 is the first one whose samples are all >= n0
    uint64_t m_main = (qb + 1 + 15) / 16 * 16 + (uint64_t)(J - 1);
    if (m_main < p.m0) m_main = p.m0 + (uint64_t)((((J - 1) % 16) - (int)(p.m0 % 16) + 16) % 16);
    const uint64_t c_max = (p.n0 + p.n - 1) / D;                           // last block that ends inside the buffer
    uint64_t m_tail = c_max + 1 < m_end ? c_max + 1 : m_end;
    if (m_tail < p.m0) m_tail = p.m0;
    if (m_main > m_tail) m_main = m_tail;
    bool edge_unit = false;
    if (m_main > p.m0) {
        const int64_t cfs = (int64_t)p.m0 - (J - 1);
        const int64_t G0 = cfs >= 0 ? cfs / 16 : -((-cfs + 15) / 16);
        const int64_t i0 = (G0 * 16 - 1) * (int64_t)D + 1;
        const int64_t i_last = (int64_t)(m_main - 1) * D;                  // last sample of the last edge output's newest block
        const int64_t len = ((i_last - i0 + 1) + 1) & ~(int64_t)1;
        if (len <= (int64_t)p.pl_edge_cap) {
            hipLaunchKernelGGL(k_pl_edge_stage, dim3((uint32_t)((len + 255) / 256), batch), dim3(256), 0, p.pre_stream ? p.pre_stream : s, q, i0, (uint32_t)len);
            if (p.pre_stream) { (void)hipEventRecord(p.pre_event, p.pre_stream); (void)hipStreamWaitEvent(s, p.pre_event, 0); }
            q.pl_edge_ms = p.m0; q.pl_edge_me = m_main;
            edge_unit = true;
        } else gen(p.m0, m_main);
    }
    gen(m_tail, m_end);
    if (m_main >= m_tail && !edge_unit) return 0;
    q.pl_m_begin = m_main; q.pl_m_end = m_tail;
    q.pl_batch = (uint32_t)batch;
    const uint64_t M = m_tail - m_main;
    const uint32_t Bn = (uint32_t)batch;
    if (J == 9) return pm_launch_main<9, 13>(q, M, Bn, edge_unit, s);   // the 1:50, 419-tap stage
    if (J > 16) {   // device-rate front ends: 42 block lags (the table is zero beyond J; J <= 48 runs the same code)
#ifndef QRL_PM_NO_K1
        // D = 25 (C2: the 25 Msps front end): the 25th phase as a K = 1 matrix instruction
        if (D == 25 && J == 42) return pm_launch_main<42, 7, QRL_PM_RP_SMALL, true>(q, M, Bn, edge_unit, s);
        if (D == 25) return pm_launch_main<48, 7, QRL_PM_RP_SMALL, true>(q, M, Bn, edge_unit, s);
#endif
        if (J == 42) switch (NS) {
            case 3: return pm_launch_main<42, 3, QRL_PM_RP_SMALL>(q, M, Bn, edge_unit, s);   case 5: return pm_launch_main<42, 5, QRL_PM_RP_SMALL>(q, M, Bn, edge_unit, s);
            case 7: return pm_launch_main<42, 7, QRL_PM_RP_SMALL>(q, M, Bn, edge_unit, s);   case 13: return pm_launch_main<42, 13, 8>(q, M, Bn, edge_unit, s);
            default: return pm_launch_main<42, 25, 16>(q, M, Bn, edge_unit, s);
        }
        switch (NS) {
            case 3: return pm_launch_main<48, 3, QRL_PM_RP_SMALL>(q, M, Bn, edge_unit, s);   case 5: return pm_launch_main<48, 5, QRL_PM_RP_SMALL>(q, M, Bn, edge_unit, s);
            case 7: return pm_launch_main<48, 7, QRL_PM_RP_SMALL>(q, M, Bn, edge_unit, s);   case 13: return pm_launch_main<48, 13, 8>(q, M, Bn, edge_unit, s);
            default: return pm_launch_main<48, 25, 16>(q, M, Bn, edge_unit, s);
        }
    }
    switch (NS) {   // other geometries: 16 block lags (the table is zero beyond J), D = 33 .. 52
    case 9: return pm_launch_main<16, 9>(q, M, Bn, edge_unit, s);    case 10: return pm_launch_main<16, 10>(q, M, Bn, edge_unit, s);
    case 11: return pm_launch_main<16, 11>(q, M, Bn, edge_unit, s);  case 12: return pm_launch_main<16, 12>(q, M, Bn, edge_unit, s);
    default: return pm_launch_main<16, 13>(q, M, Bn, edge_unit, s);
    }
}

}  // namespace qrl
