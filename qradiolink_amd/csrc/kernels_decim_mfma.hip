// kernels_decim_mfma.hip — wide decimating FIR on the f32 matrix pipe (gfx950 / CDNA4).
//
//  k_decim_mfma : rotator_cc + rational_resampler_ccf(1, D, taps), D >= 8
//                 [gr_demod_base.cpp:57,1330-1340 (front end, 1045/4181 taps);
//                  gr_demod_2fsk.cpp:82-88, gr_demod_gmsk.cpp:80-83 (per-mode first stage)]
//
// Why MFMA on an HBM-bound path: the front-end decimator costs ~42 real x complex MAC per input
// sample (167 flop / 8 B = the fp32 machine balance of MI355X), so it only stays HBM-bound if the
// FMAs run near peak.  gfx950's f32-input MFMA (v_mfma_f32_16x16x4_f32) has the SAME peak as the
// f32 VALU and is bit-for-bit a k-ordered fmaf chain, but takes ONE operand register per 1024 MACs
// instead of three per 64: the FIR stops being LDS/issue bound.  The roofline that bounds the
// kernel remains HBM; the matrix pipe is only the FMA engine.
//
// Formulation (oracle/orc_blocks.c orc_decim_fir_ccf_m16 states the same contract):
//   output m = 16a + b,  y[16a + b] = sum_u G[b][u] * x[16 a D + u],  G[b][u] = h[b D - u].
//   16x16x4 MFMA: rows = b (16 output phases of a block), cols = a (16 blocks), K = 4 values of u.
//   A operand = G[b][u0 + kk] = one ds_read_b32 of the zero-padded tap vector kept in LDS,
//   B operand = x read straight from a LINEAR tile of the input in LDS: lane (kk, a) reads sample
//   16 a D + u0 + kk (ds_read_b64 = re and im at once -> two accumulators).  The tile is padded by
//   2 samples per 16 D so the 16 block-strided lanes fall on distinct banks.
//   The u axis is cut into 4 quarters = the 4 waves of the workgroup (split-K); partial tiles meet
//   in LDS: y = (r0 + r1) + (r2 + r3).
// Pipeline: a workgroup walks `tpw` consecutive tiles of one stream.  While the matrix pipe works on
// tile t out of LDS, the 16-byte global loads of tile t+1 are already in flight into registers (all
// LDS operands use lgkmcnt, so nothing in the MFMA phase waits on vmcnt); they are rotated and
// written to LDS after the phase.  Staging is a straight copy, no transposition.
#include <algorithm>
#include <cstdlib>
#include "devmath.hpp"
#include "engine.hpp"

#ifndef QRL_MF_EXP
#define QRL_MF_EXP 0   // timing experiments only: 1 = no LDS operand reads in the MFMA loop, 2 = no MFMA (results are wrong)
#endif

namespace qrl {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// phase profile of k_decim_mfma (QRL_DBG & 32): shader-clock ticks summed over wave 0 of every workgroup
__device__ unsigned long long g_mf_prof[8];

__device__ __forceinline__ float2 mf_rot_rel(float2 x, uint32_t krel, const float2* t_hi, const float2* t_lo)
{
    return cmul_fma(x, cmul_fma(t_hi[krel >> 9], t_lo[krel & 511u]));
}

// one input sample of stream b at absolute index i (zero outside what exists so far)
__device__ __forceinline__ float2 mf_fetch(const DecimParams& P, int b, int64_t i, const float2* t_hi, uint32_t kb0,
                                           const float2* t_lo)
{
    if (i < 0) return make_float2(0.f, 0.f);
    const uint64_t ui = (uint64_t)i;
    if (ui >= P.n0 + P.n) return make_float2(0.f, 0.f);
    if (P.in) {
        if (ui >= P.n0) {
            float2 x = P.in[(size_t)b * P.in_stride + (size_t)(ui - P.n0)];
            if (P.rot_enable) x = mf_rot_rel(x, (uint32_t)(ui - P.rot_nbase - ((uint64_t)kb0 << 9)), t_hi, t_lo);
            return x;
        }
        const uint64_t d = P.n0 - ui;
        if (d > P.hist_len) return make_float2(0.f, 0.f);
        float2 hx = P.hist[(size_t)b * P.hist_len + (P.hist_len - (uint32_t)d)];
        if (P.hist_raw && P.rot_enable) {   // un-rotated history (per-channel rotators of the freq-xlating bank): exact NCO, computed directly
            const uint64_t kk = ui - P.rot_nbase;
            const float2 hi = sincos_turn(P.rot_acc + ((kk >> 9) << 9) * P.rot_inc);
            hx = cmul_fma(hx, cmul_fma(hi, P.rot_lo[(uint32_t)kk & 511u]));
        }
        return hx;
    }
    return P.in_ring.p[(size_t)b * (P.in_ring.mask + 1u) + ((uint32_t)ui & P.in_ring.mask)];
}

// the part of a tile [i_base, i_base + Jtot) that can be fetched as aligned sample PAIRS from the caller's buffer
struct TileWin { int64_t i_base; int a, k_lo, k_hi, s_lo, s_hi; };
__device__ __forceinline__ TileWin tile_window(const DecimParams& P, uint64_t mt, int Jtot, bool fast)
{
    TileWin w;
    w.i_base = (int64_t)mt * P.D - (P.nt - 1);
    w.a = (int)((w.i_base - (int64_t)P.n0) & 1);   // pairs start at j = -a so that (i - n0) is even
    w.k_lo = w.k_hi = 0;
    if (fast) {
        int64_t lo = ((int64_t)P.n0 - w.i_base + w.a + 1) >> 1;
        int64_t hi = ((int64_t)(P.n0 + P.n) - w.i_base + w.a - 1) >> 1;
        if (lo < w.a) lo = w.a;
        if (hi > ((Jtot + w.a) >> 1)) hi = (Jtot + w.a) >> 1;
        if (hi < lo) hi = lo;
        w.k_lo = (int)lo; w.k_hi = (int)hi;
    }
    w.s_lo = w.k_hi > w.k_lo ? 2 * w.k_lo - w.a : 0;
    w.s_hi = w.k_hi > w.k_lo ? 2 * w.k_hi - w.a : 0;
    return w;
}

// Prefetch registers live in the ACCUMULATOR half of the unified register file and are loaded by an asm
// statement hipcc does not count in its vmcnt bookkeeping: a compiler-visible load made hipcc split the
// destination tuples across the MFMA loop and drain vmcnt right after the issue.  tile_wait() is the
// matching explicit wait; it names every destination "+a", so no compiler copy can be scheduled between a
// load and its wait (guide 5.7 form ii; audit: no v_accvgpr_* of these registers before the wait).
template <int NLD, int NTH>
__device__ __forceinline__ void tile_issue(const DecimParams& P, int b, const TileWin& w, int tid, f32x4 (&v)[NLD])
{
    // UNCONDITIONAL loads with the pair index clamped into the caller's buffer; lanes outside
    // [k_lo, k_hi) fetch some valid pair, tile_commit sends them to the dump slots.
    // Address = stream base (SGPR pair) + 32-bit byte offset: 3 VALU per load (the host guarantees n * 8 < 4 GiB).
    const uint64_t base = reinterpret_cast<uint64_t>(P.in + (size_t)b * P.in_stride);
    const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)base), bhi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
    const uint64_t sbase = ((uint64_t)bhi << 32) | blo;
    const int q0k = (int)((w.i_base - w.a - (int64_t)P.n0) >> 1) + w.k_lo + tid;   // pair index of this thread's first load
    const int qmax = (int)(P.n >> 1) - 1;
    const int nld = __builtin_amdgcn_readfirstlane(P.nld);
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
        if (it < nld) {   // uniform: the tile needs nld <= NLD loads per thread
            const uint32_t voff = (uint32_t)min(max(q0k + NTH * it, 0), qmax) << 4;
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=a"(v[it]) : "v"(voff), "s"(sbase) : "memory");
        }
    }
}
template <int NLD>
__device__ __forceinline__ void tile_wait(f32x4 (&v)[NLD])
{
#pragma unroll
    for (int it = 0; it < NLD; ++it) asm volatile("s_waitcnt vmcnt(0)" : "+a"(v[it]) : : "memory");
}

template <int NLD, bool FAST, int NTH>
__device__ __forceinline__ void tile_commit(const DecimParams& P, int b, const TileWin& w, int tid, const f32x4 (&v)[NLD],
                                            float2* tile, int Jtot, int dump, const float2* t_hi, uint32_t kb0, const float2* t_lo)
{
    const uint32_t magic = P.magic_blk;   // ceil(2^32 / (16 D)): j / (16 D) == umulhi(j, magic) for j < 2^16
    // samples that cannot come from the aligned-pair path (history, stream edges): rare, one at a time
    for (int seg = 0; seg < 2; ++seg) {
        const int sb = seg ? w.s_hi : 0, se = seg ? Jtot : w.s_lo;
        for (int j = sb + tid; j < se; j += NTH)
            tile[j + 2 * (int)__umulhi((uint32_t)j, magic)] = mf_fetch(P, b, w.i_base + j, t_hi, kb0, t_lo);
    }
    if (!FAST) return;
    // Interior tiles (all of them except the first / last of a stream): every pair of the tile comes from the caller's
    // buffer, pairs are aligned to the tile (a = 0) and to the rotator tables (even krel).  Then a thread's two samples
    // are LDS neighbours (the 2-sample pads sit at even positions), its fine-table factors are one 16-byte LDS read, its
    // coarse factors are t_hi[h0 + it] -- fetched up front so no LDS latency sits inside the loop -- and pairs past the
    // end of the tile simply land in the 512-sample slack behind it: no predication, one ds_write_b128 per pair.
    if constexpr (NTH == 256 && NLD <= 16) {
        const uint64_t kbase = (uint64_t)w.i_base - P.rot_nbase - ((uint64_t)kb0 << 9);
        const int nld = __builtin_amdgcn_readfirstlane(P.nld);
        if (w.a == 0 && w.k_lo == 0 && w.k_hi == (Jtot >> 1) && !(kbase & 1u) && P.rot_enable && nld <= NLD) {
            const int j0 = 2 * tid;
            const uint32_t krel0 = (uint32_t)kbase + (uint32_t)j0;
            const float4 lo01 = *reinterpret_cast<const float4*>(t_lo + (krel0 & 511u));
            const float2 lo0 = make_float2(lo01.x, lo01.y), lo1 = make_float2(lo01.z, lo01.w);
            const float2* hp0 = t_hi + (krel0 >> 9);
            float2 hi[NLD];
#pragma unroll
            for (int it = 0; it < NLD; ++it) hi[it] = hp0[it];
#pragma unroll
            for (int it = 0; it < NLD; ++it) {
                if (it < nld) {
                    const int j = j0 + 512 * it;
                    float2 x0 = make_float2(v[it].x, v[it].y), x1 = make_float2(v[it].z, v[it].w);
                    x0 = cmul_fma(x0, cmul_fma(hi[it], lo0));
                    x1 = cmul_fma(x1, cmul_fma(hi[it], lo1));
                    const int p0 = j + 2 * (int)__umulhi((uint32_t)j, magic);
                    *reinterpret_cast<float4*>(tile + p0) = make_float4(x0.x, x0.y, x1.x, x1.y);
                }
            }
            return;
        }
    }
    // register-prefetched pairs: STRAIGHT-LINE code (no exec branches, so hipcc batches the LDS reads);
    // pairs past the end of the window are written to a per-thread dump slot behind the tile
    const int j0 = 2 * (w.k_lo + tid) - w.a;
    const uint32_t krel0 = (uint32_t)((uint64_t)(w.i_base + j0) - P.rot_nbase - ((uint64_t)kb0 << 9));
    // krel advances by 512 per step: the fine-table factors of a thread never change
    const float2 lo0 = t_lo[krel0 & 511u], lo1 = t_lo[(krel0 + 1u) & 511u];
    const int npair = w.k_hi - w.k_lo - tid;   // this thread owns pairs it < ceil(npair / NTH)
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
        const int j = j0 + 2 * NTH * it;
        const uint32_t krel = krel0 + 2u * NTH * it;
        float2 x0 = make_float2(v[it].x, v[it].y), x1 = make_float2(v[it].z, v[it].w);
        x0 = cmul_fma(x0, cmul_fma(t_hi[krel >> 9], lo0));   // the caller-buffer path always carries the rotator
        x1 = cmul_fma(x1, cmul_fma(t_hi[(krel + 1u) >> 9], lo1));
        const bool ok = NTH * it < npair;
        const int p0 = ok ? j + 2 * (int)__umulhi((uint32_t)j, magic) : dump + 2 * tid;
        const int p1 = ok ? j + 1 + 2 * (int)__umulhi((uint32_t)(j + 1), magic) : dump + 2 * tid + 1;
        tile[p0] = x0;
        tile[p1] = x1;
    }
}

// One contiguous piece of the step loop (no pad jump inside): step i reads A = ap[-4 i] and B = bp[BS i].
// Software pipelined by hand with two register sets (X = chunk c, Y = chunk c + 1).  The LDS reads are asm
// statements: hipcc would fuse neighbouring ds_read_b64 into ds_read2_b64, which is served at HALF the LDS
// rate with mod-32 banking (MI355X_MICROARCH.md, LDS table) and made the LDS the bottleneck of the CU.
// Waits are explicit and name their registers (guide 5.7 form ii):
//   issue Y.b (8 DS) | lgkmcnt(8): X complete | 8 MFMA | issue Y.a (8 DS) | 8 MFMA
constexpr int MF_U = 8;
typedef __attribute__((address_space(3))) const void* lds_cptr;
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(lds_cptr)p; }

#define MF_RD64(dst, addr, off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
#define MF_RD32(dst, addr, off) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))

struct MfSet2 { float a[MF_U]; float2 b[MF_U]; };   // NA = 16: B = one complex sample per lane
struct MfSet1 { float a[MF_U]; float b[MF_U]; };    // NA = 8 : B = re or im of a sample

// B reads of 8 steps: byte stride BSB between steps
template <int BSB>
__device__ __forceinline__ void mf_issue_b(MfSet2& r, uint32_t baddr)
{
    MF_RD64(r.b[0], baddr, 0 * BSB); MF_RD64(r.b[1], baddr, 1 * BSB); MF_RD64(r.b[2], baddr, 2 * BSB); MF_RD64(r.b[3], baddr, 3 * BSB);
    MF_RD64(r.b[4], baddr, 4 * BSB); MF_RD64(r.b[5], baddr, 5 * BSB); MF_RD64(r.b[6], baddr, 6 * BSB); MF_RD64(r.b[7], baddr, 7 * BSB);
}
template <int BSB>
__device__ __forceinline__ void mf_issue_b(MfSet1& r, uint32_t baddr)
{
    MF_RD32(r.b[0], baddr, 0 * BSB); MF_RD32(r.b[1], baddr, 1 * BSB); MF_RD32(r.b[2], baddr, 2 * BSB); MF_RD32(r.b[3], baddr, 3 * BSB);
    MF_RD32(r.b[4], baddr, 4 * BSB); MF_RD32(r.b[5], baddr, 5 * BSB); MF_RD32(r.b[6], baddr, 6 * BSB); MF_RD32(r.b[7], baddr, 7 * BSB);
}
// A reads of 8 steps: step u at aaddr7 + 16 (7 - u) bytes (aaddr7 = address of the LAST step of the chunk)
template <class SET>
__device__ __forceinline__ void mf_issue_a(SET& r, uint32_t aaddr7)
{
    MF_RD32(r.a[0], aaddr7, 112); MF_RD32(r.a[1], aaddr7, 96); MF_RD32(r.a[2], aaddr7, 80); MF_RD32(r.a[3], aaddr7, 64);
    MF_RD32(r.a[4], aaddr7, 48);  MF_RD32(r.a[5], aaddr7, 32); MF_RD32(r.a[6], aaddr7, 16); MF_RD32(r.a[7], aaddr7, 0);
}
__device__ __forceinline__ void mf_wait8(MfSet2& r)
{
    asm volatile("s_waitcnt lgkmcnt(8)"
                 : "+v"(r.a[0]), "+v"(r.a[1]), "+v"(r.a[2]), "+v"(r.a[3]), "+v"(r.a[4]), "+v"(r.a[5]), "+v"(r.a[6]), "+v"(r.a[7]),
                   "+v"(r.b[0]), "+v"(r.b[1]), "+v"(r.b[2]), "+v"(r.b[3]), "+v"(r.b[4]), "+v"(r.b[5]), "+v"(r.b[6]), "+v"(r.b[7]));
}
__device__ __forceinline__ void mf_wait8(MfSet1& r)
{
    asm volatile("s_waitcnt lgkmcnt(8)"
                 : "+v"(r.a[0]), "+v"(r.a[1]), "+v"(r.a[2]), "+v"(r.a[3]), "+v"(r.a[4]), "+v"(r.a[5]), "+v"(r.a[6]), "+v"(r.a[7]),
                   "+v"(r.b[0]), "+v"(r.b[1]), "+v"(r.b[2]), "+v"(r.b[3]), "+v"(r.b[4]), "+v"(r.b[5]), "+v"(r.b[6]), "+v"(r.b[7]));
}
__device__ __forceinline__ void mf_wait0(MfSet2& r)
{
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(r.a[0]), "+v"(r.a[1]), "+v"(r.a[2]), "+v"(r.a[3]), "+v"(r.a[4]), "+v"(r.a[5]), "+v"(r.a[6]), "+v"(r.a[7]),
                   "+v"(r.b[0]), "+v"(r.b[1]), "+v"(r.b[2]), "+v"(r.b[3]), "+v"(r.b[4]), "+v"(r.b[5]), "+v"(r.b[6]), "+v"(r.b[7]));
}
__device__ __forceinline__ void mf_wait0(MfSet1& r)
{
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(r.a[0]), "+v"(r.a[1]), "+v"(r.a[2]), "+v"(r.a[3]), "+v"(r.a[4]), "+v"(r.a[5]), "+v"(r.a[6]), "+v"(r.a[7]),
                   "+v"(r.b[0]), "+v"(r.b[1]), "+v"(r.b[2]), "+v"(r.b[3]), "+v"(r.b[4]), "+v"(r.b[5]), "+v"(r.b[6]), "+v"(r.b[7]));
}
template <int U0>
__device__ __forceinline__ void mf_fma4(const MfSet2& c, f32x4& acc0, f32x4& acc1)
{
#pragma unroll
    for (int u = U0; u < U0 + 4; ++u) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(c.a[u], c.b[u].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(c.a[u], c.b[u].y, acc1, 0, 0, 0);
    }
}
template <int U0>
__device__ __forceinline__ void mf_fma4(const MfSet1& c, f32x4& acc0, f32x4&)
{
#pragma unroll
    for (int u = U0; u < U0 + 4; ++u) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(c.a[u], c.b[u], acc0, 0, 0, 0);
}

// SET/BT/BS: MfSet2/float2/4 (B stride 4 float2 = 32 B per step) or MfSet1/float/8 (8 floats = 32 B per step)
template <class SET, typename BT, int BS>
__device__ __forceinline__ void mfma_piece(const float* __restrict__ ap, const BT* __restrict__ bp, int n, f32x4& acc0, f32x4& acc1)
{
    constexpr int U = MF_U;
    constexpr int BSB = 32;   // bytes between the B operands of consecutive steps
    int i = 0;
    if (n >= U) {
        const uint32_t a0 = lds_addr(ap) - 16u * (U - 1);   // address of step 7 of chunk 0; chunk c: minus 128 c
        const uint32_t b0 = lds_addr(bp);                   // chunk c: plus 256 c
        const int nch = n / U;
        SET x, y;
        mf_issue_b<BSB>(x, b0);
        mf_issue_a(x, a0);
        for (int c = 0; c < nch; c += 2) {
            // chunk c lives in x; prefetch chunk c + 1 into y (beyond the end: harmless re-read of chunk 0)
            const int c1 = c + 1 < nch ? c + 1 : 0;
            mf_issue_b<BSB>(y, b0 + 256u * c1);
            mf_wait8(x);
            __builtin_amdgcn_sched_barrier(0);
            mf_fma4<0>(x, acc0, acc1);
            __builtin_amdgcn_sched_barrier(0);
            mf_issue_a(y, a0 - 128u * c1);
            __builtin_amdgcn_sched_barrier(0);
            mf_fma4<4>(x, acc0, acc1);
            __builtin_amdgcn_sched_barrier(0);
            if (c + 1 >= nch) { mf_wait0(y); break; }
            const int c2 = c + 2 < nch ? c + 2 : 0;
            mf_issue_b<BSB>(x, b0 + 256u * c2);
            mf_wait8(y);
            __builtin_amdgcn_sched_barrier(0);
            mf_fma4<0>(y, acc0, acc1);
            __builtin_amdgcn_sched_barrier(0);
            mf_issue_a(x, a0 - 128u * c2);
            __builtin_amdgcn_sched_barrier(0);
            mf_fma4<4>(y, acc0, acc1);
            __builtin_amdgcn_sched_barrier(0);
            if (c + 2 >= nch) { mf_wait0(x); break; }
        }
        i = nch * U;
    }
    for (; i < n; ++i) {
        const float av = ap[-4 * i];
        const BT bv = bp[BS * i];
        if constexpr (BS == 4) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, reinterpret_cast<const float2&>(bv).x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, reinterpret_cast<const float2&>(bv).y, acc1, 0, 0, 0);
        } else {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, reinterpret_cast<const float&>(bv), acc0, 0, 0, 0);
        }
    }
}

// Compiler-scheduled variant of the same ping-pong (plain LDS reads, sched_barrier pinned).  It is the DEFAULT:
// the asm variant above is ~5 % faster but showed rare (1e-4 per tile) wrong outputs with two workgroups per
// CU at 25 Msps that could not be explained; build with -DQRL_MF_ASM_LDS=1 to select it for experiments.
#ifndef QRL_MF_INTERLEAVE
#define QRL_MF_INTERLEAVE 1
#endif
#ifndef QRL_MF_VOLATILE_LDS
#define QRL_MF_VOLATILE_LDS 1
#endif
#ifndef QRL_MF_ASM_LDS
#define QRL_MF_ASM_LDS 0
#endif
template <typename BT, int BS>
struct MfChunk {
    float a[MF_U]; BT b[MF_U];
    __device__ __forceinline__ void load(const float* ap, const BT* bp, int i)
    {
#pragma unroll
        for (int u = 0; u < MF_U; ++u) {
#if QRL_MF_VOLATILE_LDS
            // volatile LDS-address-space reads: hipcc must not fuse neighbours into ds_read2_b32 / ds_read2_b64, which the
            // LDS serves at half rate (MI355X_MICROARCH.md, LDS table).  The compiler still owns the waitcnt bookkeeping
            // (unlike the asm variant above).  Measured: front end 2.10 -> 2.02 ms (C2), 9.65 -> 9.49 ms (C1).
            typedef float f32x2_t __attribute__((ext_vector_type(2)));
            typedef const volatile __attribute__((address_space(3))) float* lds_vf;
            typedef const volatile __attribute__((address_space(3))) f32x2_t* lds_vf2;
            a[u] = *(lds_vf)(lds_cptr)(ap - 4 * (i + u));
            if constexpr (sizeof(BT) == 8) {
                const f32x2_t t = *(lds_vf2)(lds_cptr)(bp + BS * (i + u));
                b[u] = BT{t.x, t.y};
            } else {
                b[u] = *(lds_vf)(lds_cptr)(bp + BS * (i + u));
            }
#else
            a[u] = ap[-4 * (i + u)]; b[u] = bp[BS * (i + u)];
#endif
        }
    }
};
#ifndef QRL_MF_EXP
#define QRL_MF_EXP 0
#endif
__device__ __forceinline__ void mf_fma(const MfChunk<float2, 4>& c, f32x4& acc0, f32x4& acc1)
{
#if QRL_MF_EXP == 2
#pragma unroll
    for (int u = 0; u < MF_U; ++u) { acc0[0] = fmaf(c.a[u], c.b[u].x, acc0[0]); acc1[0] = fmaf(c.a[u], c.b[u].y, acc1[0]); }
    return;
#endif
#pragma unroll
    for (int u = 0; u < MF_U; ++u) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(c.a[u], c.b[u].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(c.a[u], c.b[u].y, acc1, 0, 0, 0);
    }
}
__device__ __forceinline__ void mf_fma(const MfChunk<float, 8>& c, f32x4& acc0, f32x4&)
{
#pragma unroll
    for (int u = 0; u < MF_U; ++u) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(c.a[u], c.b[u], acc0, 0, 0, 0);
}
template <typename BT, int BS>
__device__ __forceinline__ void mfma_piece_c(const float* __restrict__ ap, const BT* __restrict__ bp, int n, f32x4& acc0, f32x4& acc1)
{
    constexpr int U = MF_U;
    int i = 0;
    if (n >= U) {
        MfChunk<BT, BS> r0, r1;
        r0.load(ap, bp, 0);
        while (i + 2 * U <= n) {
#if QRL_MF_EXP == 1
            if (i == 0)
#endif
            r1.load(ap, bp, i + U);
#if !QRL_MF_INTERLEAVE
            __builtin_amdgcn_sched_barrier(0);
#endif
            mf_fma(r0, acc0, acc1);
#if QRL_MF_INTERLEAVE
            // the wave issues in order and blocks at an MFMA while the pipe is busy (32 cycles per v_mfma_f32_16x16x4_f32, one
            // dependent chain already runs at that rate: tools/ubench/mfma_chain.hip); everything issued BETWEEN two MFMAs is
            // free, a block of operand reads behind 16 MFMAs is not.  Interleave: one LDS read group after every MFMA.
            {
                constexpr int NM = BS == 4 ? 2 * U : U, RD = 2 * U / NM;   // MFMAs per chunk; operand reads per MFMA
#pragma unroll
                for (int k = 0; k < NM; ++k) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, RD, 0); }
            }
            __builtin_amdgcn_sched_barrier(0);
#else
            __builtin_amdgcn_sched_barrier(0);
#endif
#if QRL_MF_EXP != 1
            r0.load(ap, bp, i + 3 * U <= n ? i + 2 * U : 0);   // beyond the end: harmless re-read of chunk 0
#endif
#if !QRL_MF_INTERLEAVE
            __builtin_amdgcn_sched_barrier(0);
#endif
            mf_fma(r1, acc0, acc1);
#if QRL_MF_INTERLEAVE
            {
                constexpr int NM = BS == 4 ? 2 * U : U, RD = 2 * U / NM;
#pragma unroll
                for (int k = 0; k < NM; ++k) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 1); __builtin_amdgcn_sched_group_barrier(0x100, RD, 1); }
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
            i += 2 * U;
        }
        if (i + U <= n) { mf_fma(r0, acc0, acc1); i += U; }
    }
    for (; i < n; ++i) {
        MfChunk<BT, BS> r;   // single step
        r.a[0] = ap[-4 * i]; r.b[0] = bp[BS * i];
        if constexpr (BS == 4) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(r.a[0], reinterpret_cast<const float2&>(r.b[0]).x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(r.a[0], reinterpret_cast<const float2&>(r.b[0]).y, acc1, 0, 0, 0);
        } else {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(r.a[0], reinterpret_cast<const float&>(r.b[0]), acc0, 0, 0, 0);
        }
    }
}


// ---- one wave's quarter of the FIR of one tile on the matrix pipe; partial sums -> part[g][.] ----
template <int NA, bool ALIAS>
__device__ __forceinline__ void mfma_quarter(const float2* tile, const float* hp, float2* part, int g, int lane, int D, int S,
                                             uint32_t magic_seg)
{
    constexpr int T = 16 * NA;
    const int Sq = S >> 2, Pp = 16 * D + 2, kk = lane >> 4, seg_len = 4 * D;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    int s = g * Sq;
    const int s_end = s + Sq;
    if constexpr (NA == 16) {
        const int acol = lane & 15;
        const float2* bp = tile + acol * Pp + kk;
        const float* ap = hp + (acol * D + 4 * S - kk);     // tap k = b D - u, stored at k + (4 S - nt + 1)
        while (s < s_end) {
            const int seg = (int)__umulhi((uint32_t)s, magic_seg);
            const int e = min(s_end, (seg + 1) * seg_len);
            if constexpr (QRL_MF_ASM_LDS) mfma_piece<MfSet2, float2, 4>(ap - 4 * s, bp + 4 * s + 2 * seg, e - s, acc0, acc1);
            else mfma_piece_c<float2, 4>(ap - 4 * s, bp + 4 * s + 2 * seg, e - s, acc0, acc1);
            s = e;
        }
        if constexpr (ALIAS) __syncthreads();   // part aliases the head of the tile: every wave must be done reading it
        float2* pp = part + g * T + 16 * acol + 4 * kk;
#pragma unroll
        for (int r = 0; r < 4; ++r) pp[r] = make_float2(acc0[r], acc1[r]);
    } else {
        // NA = 8: columns = 8 blocks x {re, im};  NA = 4: 4 blocks x {re, im} x 2 (the odd columns duplicate the even
        // ones and are dropped: half the matrix pipe is wasted, which is irrelevant where this variant is used --
        // low-rate-ratio decimators whose MFMA time is a few percent -- and buys a tile small enough for 3 workgroups per CU)
        constexpr int SH = NA == 8 ? 1 : 2;
        const int n = lane & 15, acol = n >> SH, c = (n >> (SH - 1)) & 1;
        const bool keep = NA == 8 || !(n & 1);
        const float* bp = reinterpret_cast<const float*>(tile) + 2 * (acol * Pp + kk) + c;
        const float* ap = hp + ((lane & 15) * D + 4 * S - kk);
        while (s < s_end) {
            const int seg = (int)__umulhi((uint32_t)s, magic_seg);
            const int e = min(s_end, (seg + 1) * seg_len);
            if constexpr (QRL_MF_ASM_LDS) mfma_piece<MfSet1, float, 8>(ap - 4 * s, bp + 2 * (4 * s + 2 * seg), e - s, acc0, acc1);
            else mfma_piece_c<float, 8>(ap - 4 * s, bp + 2 * (4 * s + 2 * seg), e - s, acc0, acc1);
            s = e;
        }
        if constexpr (ALIAS) __syncthreads();
        float* pf = reinterpret_cast<float*>(part + g * T + 16 * acol + 4 * kk) + c;
        if (keep) {
#pragma unroll
            for (int r = 0; r < 4; ++r) pf[2 * r] = acc0[r];
        }
    }
}

// ---- one-team variant (tiles too large for two LDS buffers, or input from an engine ring): a workgroup
// of 4 waves walks its tiles; the loads of tile k + 1 fly during the MFMA phase of tile k.
template <int NA, int NLD, bool FAST, bool ALIAS, int WPE, int NTH>
__global__ __launch_bounds__(NTH) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_decim_mfma(const DecimParams P_)
{
    const DecimParams& P = P_;
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int T = 16 * NA;
    const int D = P.D, S = P.S;
    const int blk = 16 * D;
    const int Jtot = (NA - 1) * blk + 4 * S;
    const int hpn = (15 * D + 4 * S + 4 + 3) & ~3;   // multiple of 4 floats: the tile behind it stays 16-byte aligned
    const int nhi = (P.nhi + 1) & ~1;
    float2* t_lo = reinterpret_cast<float2*>(smem);            // 512
    float2* t_hi = t_lo + 512;                                 // nhi
    float* hp = reinterpret_cast<float*>(t_hi + nhi);          // zero-padded taps
    float2* tile = reinterpret_cast<float2*>(hp + hpn);
    const int dump0 = Jtot + 2 * (Jtot / blk) + 4;
    float2* part = ALIAS ? tile : tile + dump0 + 2 * NTH;          // 4 T partial sums (ALIAS: on the head of the tile)
    const int dump = Jtot + 2 * (Jtot / blk) + 4;              // 512 dump slots behind the tile (tile_commit)

    const int b = blockIdx.y;
    // grid.x is either a multiple of 8 (then neighbouring chunks are mapped to the same XCD/L2: block b runs on
    // XCD b % 8) or exactly nchunks (few chunks per stream: the stream index spreads the work over the XCDs)
    const uint32_t per = gridDim.x >> 3;
    const uint32_t cix = (gridDim.x & 7u) ? blockIdx.x : (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    // workgroup cix walks tiles cix, cix + nchunks, ...: the workgroups resident at any moment sweep ONE
    // contiguous region of the stream together (halos shared in L2)
    const uint32_t nchunks = P.nchunks;
    if (cix >= nchunks || cix >= P.tiles) return;
    const int tid = threadIdx.x;
    const uint64_t mt_first = (P.m0 / T) * (uint64_t)T;

    for (int k = tid; k < hpn; k += NTH) hp[k] = P.gtab[k];
    if (P.rot_enable) { for (int k = tid; k < 512; k += NTH) t_lo[k] = P.rot_lo[k]; }
    const int g = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;

    f32x4 v[NLD];
    TileWin w = tile_window(P, mt_first + (uint64_t)cix * T, Jtot, FAST);
    if constexpr (FAST) tile_issue<NLD, NTH>(P, b, w, tid, v);
    const bool prof = (P.dbg & 32) && tid == 0;
    unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = prof ? __builtin_readcyclecounter() : 0;
#define MF_STAMP(k) do { if (prof) { const unsigned long long tn_ = __builtin_readcyclecounter(); pc[k] += tn_ - tprev; tprev = tn_; } } while (0)
    for (uint32_t t = cix; t < P.tiles; t += nchunks) {
        const uint64_t mt = mt_first + (uint64_t)t * T;
        uint32_t kb0 = 0;
        if (P.rot_enable) {   // coarse rotator table of this tile
            const int64_t first_new = w.i_base > (int64_t)P.n0 ? w.i_base : (int64_t)P.n0;
            kb0 = (uint32_t)(((uint64_t)first_new - P.rot_nbase) >> 9);
            if (tid < P.nhi) t_hi[tid] = sincos_turn(P.rot_acc + ((uint64_t)(kb0 + tid) << 9) * P.rot_inc);
        }
        __syncthreads();
        MF_STAMP(0);
        if constexpr (FAST) tile_wait<NLD>(v);
        MF_STAMP(1);
        tile_commit<NLD, FAST, NTH>(P, b, w, tid, v, tile, Jtot, dump, t_hi, kb0, t_lo);
        MF_STAMP(2);
        __syncthreads();
        MF_STAMP(3);
        if (t + nchunks < P.tiles) {
            w = tile_window(P, mt + (uint64_t)nchunks * T, Jtot, FAST);
            if constexpr (FAST) tile_issue<NLD, NTH>(P, b, w, tid, v);
        }
        MF_STAMP(4);
        // waves 0-3 = the four quarters of the contract; with 512 threads waves 4-7 only help staging (more waves per
        // SIMD hide the LDS / dependency latency of the commit) and wait at the barriers of the quarter function
        if (NTH == 256 || g < 4) mfma_quarter<NA, ALIAS>(tile, hp, part, g, lane, D, S, P.magic_seg);
        else if (ALIAS) __syncthreads();
        MF_STAMP(5);
        __syncthreads();
        if (tid < T) {
            const uint64_t m = mt + tid;
            if (m >= P.m0 && m < P.m0 + P.m_count) {
                const float2 r0 = part[tid], r1 = part[T + tid], r2 = part[2 * T + tid], r3 = part[3 * T + tid];
                float2 y;
                y.x = (r0.x + r1.x) + (r2.x + r3.x);
                y.y = (r0.y + r1.y) + (r2.y + r3.y);
                P.out.p[((size_t)b * (P.out_row_mul_m1 + 1u) + P.out_row_add) * (P.out.mask + 1u) + ((uint32_t)m & P.out.mask)] = y;
            }
        }
        MF_STAMP(6);
    }
    if (prof) {
#pragma unroll
        for (int k = 0; k < 7; ++k) atomicAdd(&g_mf_prof[k], pc[k]);
        atomicAdd(&g_mf_prof[7], 1ull);
    }
#undef MF_STAMP
}

// ---- two-team variant: ONE workgroup of 8 waves per CU, two tile buffers.  In every phase one team
// (4 waves = the 4 quarters) runs the MFMA loop of its tile while the other team combines its previous
// tile, commits its next tile into its own buffer and issues the loads of the one after: the VALU/LDS
// staging work of one team is deterministically overlapped with the matrix-pipe work of the other
// (two independent workgroups per CU drift into phase and serialise instead).  One s_barrier per phase.
template <int NA, int NLD>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_decim_mfma2(const DecimParams P_)
{
    const DecimParams& P = P_;
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int T = 16 * NA;
    const int D = P.D, S = P.S;
    const int blk = 16 * D;
    const int Jtot = (NA - 1) * blk + 4 * S;
    const int hpn = (15 * D + 4 * S + 4 + 3) & ~3;
    const int nhi = (P.nhi + 1) & ~1;
    const int dump = Jtot + 2 * (Jtot / blk) + 4;
    const int tile_len = (dump + 512 + 16 + 1) & ~1;
    float2* t_lo = reinterpret_cast<float2*>(smem);            // 512
    float2* t_hi_all = t_lo + 512;                             // [team][buf][nhi]
    float* hp = reinterpret_cast<float*>(t_hi_all + 4 * nhi);  // hpn floats
    float2* part_all = reinterpret_cast<float2*>(hp + hpn);    // [team][4 T]
    float2* tile_all = part_all + 2 * 4 * T;                   // [team][tile_len]

    const int b = blockIdx.y;
    const uint32_t per = gridDim.x >> 3;
    const uint32_t cix = (gridDim.x & 7u) ? blockIdx.x : (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    const uint32_t nchunks = P.nchunks;
    if (cix >= nchunks || cix >= P.tiles) return;
    const int tid = threadIdx.x;
    const int team = __builtin_amdgcn_readfirstlane(tid >> 8);
    const int ttid = tid & 255;
    const int g = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);
    const int lane = tid & 63;
    const uint64_t mt_first = (P.m0 / T) * (uint64_t)T;
    const int K = (int)((P.tiles - cix + nchunks - 1) / nchunks);   // tiles of this workgroup: t(k) = cix + k nchunks
    float2* tile = tile_all + (size_t)team * tile_len;
    float2* part = part_all + (size_t)team * 4 * T;
    float2* t_hi0 = t_hi_all + (size_t)team * 2 * nhi;

    for (int k = tid; k < hpn; k += 512) hp[k] = P.gtab[k];
    if (P.rot_enable) t_lo[tid] = P.rot_lo[tid];

    f32x4 v[NLD];
    TileWin w;
    uint32_t kb0 = 0;
    auto tile_mt = [&](int k) { return mt_first + ((uint64_t)cix + (uint64_t)k * nchunks) * T; };
    // coarse rotator table of tile k into buffer (k >> 1) & 1 of this team; returns its kb0
    auto make_thi = [&](int k) -> uint32_t {
        const int64_t ib = (int64_t)tile_mt(k) * D - (P.nt - 1);
        const int64_t first_new = ib > (int64_t)P.n0 ? ib : (int64_t)P.n0;
        const uint32_t kb = (uint32_t)(((uint64_t)first_new - P.rot_nbase) >> 9);
        if (P.rot_enable && ttid < P.nhi)
            t_hi0[((k >> 1) & 1) * nhi + ttid] = sincos_turn(P.rot_acc + ((uint64_t)(kb + ttid) << 9) * P.rot_inc);
        return kb;
    };
    // stage role: (combine of tile kc done by the caller) -> wait + commit tile k -> issue loads of tile k + 2 -> table of k + 2
    const bool prof = (P.dbg & 32) && ttid == 0;
    unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = 0;
#define MF2_T0() do { if (prof) tprev = __builtin_readcyclecounter(); } while (0)
#define MF2_STAMP(k) do { if (prof) { const unsigned long long tn_ = __builtin_readcyclecounter(); pc[k] += tn_ - tprev; tprev = tn_; } } while (0)
    auto stage = [&](int k) {
        MF2_T0();
        tile_wait<NLD>(v);
        MF2_STAMP(0);
        tile_commit<NLD, true, 256>(P, b, w, ttid, v, tile, Jtot, dump, t_hi0 + ((k >> 1) & 1) * nhi, kb0, t_lo);
        MF2_STAMP(1);
        if (k + 2 < K) {
            w = tile_window(P, tile_mt(k + 2), Jtot, true);
            tile_issue<NLD, 256>(P, b, w, ttid, v);
            kb0 = make_thi(k + 2);
        }
        MF2_STAMP(2);
    };
    auto combine = [&](int k) {
        if (ttid < T) {
            const uint64_t m = tile_mt(k) + ttid;
            if (m >= P.m0 && m < P.m0 + P.m_count) {
                const float2 r0 = part[ttid], r1 = part[T + ttid], r2 = part[2 * T + ttid], r3 = part[3 * T + ttid];
                float2 y;
                y.x = (r0.x + r1.x) + (r2.x + r3.x);
                y.y = (r0.y + r1.y) + (r2.y + r3.y);
                P.out.p[((size_t)b * (P.out_row_mul_m1 + 1u) + P.out_row_add) * (P.out.mask + 1u) + ((uint32_t)m & P.out.mask)] = y;
            }
        }
    };

    // start-up: each team issues the loads and the table of its first tile (k = team)
    if (team < K) {
        w = tile_window(P, tile_mt(team), Jtot, true);
        tile_issue<NLD, 256>(P, b, w, ttid, v);
        kb0 = make_thi(team);
    }
    __syncthreads();
    if (team == 0) stage(0);
    __syncthreads();
    // phase p: team (p & 1) runs the matrix pipe on tile p, the other team combines tile p - 1 and stages tile p + 1
    for (int p = 0; p <= K; ++p) {
        if ((p & 1) == team) {
            MF2_T0();
            if (p < K) mfma_quarter<NA, false>(tile, hp, part, g, lane, D, S, P.magic_seg);
            MF2_STAMP(3);
        } else {
            MF2_T0();
            if (p >= 1) combine(p - 1);
            MF2_STAMP(4);
            if (p + 1 < K) stage(p + 1);
        }
        MF2_T0();
        __syncthreads();
        MF2_STAMP(5 + ((p & 1) == team ? 0 : 1));
    }
    if (prof) {
#pragma unroll
        for (int k = 0; k < 7; ++k) atomicAdd(&g_mf_prof[k], pc[k]);
        atomicAdd(&g_mf_prof[7], 1ull);
    }
}

void decim_mfma_prof_read(unsigned long long* out8)
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_mf_prof), 8 * sizeof(unsigned long long));
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_mf_prof), z, sizeof z);
}
int decim_mfma_steps(int nt, int D)
{
    const int S = (nt + 15 * D + 3) / 4;
    return (S + 3) / 4 * 4;
}
int decim_mfma_hpn(int nt, int D) { return (15 * D + 4 * decim_mfma_steps(nt, D) + 4 + 3) & ~3; }
static long long mfma_jtot(int nt, int D, int NA) { return (long long)(NA - 1) * 16 * D + 4LL * decim_mfma_steps(nt, D); }
static int mfma_nhi(int nt, int D, int NA, int) { return (int)(mfma_jtot(nt, D, NA) / 512 + 3); }
static size_t mfma_lds(int nt, int D, int NA, int tpw)
{
    const long long Jtot = mfma_jtot(nt, D, NA);
    const long long npos = Jtot + 2 * (Jtot / (16LL * D)) + 4 + 512 + 16;   // + dump slots (+ pad slack of the unpredicated commit)
    return (size_t)(512 + ((mfma_nhi(nt, D, NA, tpw) + 1) & ~1) + npos) * sizeof(float2) + (size_t)decim_mfma_hpn(nt, D) * sizeof(float);
}
// rule shared with oracle/orc_blocks.c orc_decim_uses_m16
bool decim_uses_mfma(int nt, int D)
{
    if (D < 8) return false;
    const long long samples = 7LL * 16 * D + 4LL * decim_mfma_steps(nt, D);
    return (samples + 2 * (samples / (16LL * D)) + 64) * 8 <= 150 * 1024;
}
constexpr int kTpwMax = 16;
static size_t mfma2_lds(int nt, int D, int NA);
// 16 output blocks per tile when two workgroups of that size fit the 160 KB of a CU, else 8
int decim_mfma_na(int nt, int D)
{
    if (const char* e = std::getenv("QRL_DECIM_NA")) { const int v = std::atoi(e); if (v == 4 || v == 8 || v == 16) return v; }   // experiments
    if (mfma_lds(nt, D, 16, kTpwMax) <= 80 * 1024) return 16;    // two (or more) workgroups per CU with the efficient tile
    if (mfma_lds(nt, D, 8, kTpwMax) <= 160 * 1024) return 8;   // (4-block tiles with three workgroups per CU were measured slower)
    return 4;   // very long filters (100:1 front end, 4181 taps): only the 4-block tile fits the 160 KB of a CU
}
size_t decim_mfma_lds_bytes(int nt, int D) { return mfma_lds(nt, D, decim_mfma_na(nt, D), kTpwMax); }

template <int NA, int NLD, bool FAST, int WPE = 2, int NTH = 256>
static void launch_k(const DecimParams& q, dim3 grid, size_t lds, hipStream_t s)
{
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_decim_mfma<NA, NLD, FAST, true, WPE, NTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_decim_mfma<NA, NLD, FAST, false, WPE, NTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    const char* na = std::getenv("QRL_DECIM_NOALIAS");
    if (na && na[0] == '1') hipLaunchKernelGGL((k_decim_mfma<NA, NLD, FAST, false, WPE, NTH>), grid, dim3(NTH), lds + 4 * 16 * NA * sizeof(float2) + (NTH - 256) * 2 * sizeof(float2), s, q);
    else {
        const char* pad = std::getenv("QRL_DECIM_PADLDS");   // debugging aid: extra LDS to force one workgroup per CU
        hipLaunchKernelGGL((k_decim_mfma<NA, NLD, FAST, true, WPE, NTH>), grid, dim3(NTH), lds + (pad ? std::atoi(pad) : 0) + (NTH - 256) * 2 * sizeof(float2), s, q);
    }
}
static size_t mfma2_lds(int nt, int D, int NA)
{
    const long long Jtot = mfma_jtot(nt, D, NA);
    const long long dump = Jtot + 2 * (Jtot / (16LL * D)) + 4;
    const long long tile_len = (dump + 512 + 16 + 1) & ~1LL;
    const long long nhi = (mfma_nhi(nt, D, NA, 1) + 1) & ~1;
    return (size_t)(512 + 4 * nhi + 2 * 4 * 16 * NA + 2 * tile_len) * sizeof(float2) + (size_t)decim_mfma_hpn(nt, D) * sizeof(float);
}
template <int NA, int NLD>
static void launch_k2(const DecimParams& q, dim3 grid, size_t lds, hipStream_t s)
{
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_decim_mfma2<NA, NLD>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL((k_decim_mfma2<NA, NLD>), grid, dim3(512), lds, s, q);
}
template <int NA, int NLD>
static void launch_one(const DecimParams& q, dim3 grid, size_t lds, hipStream_t s)
{
    // two-team kernel: caller's buffer, even-aligned fast path, two tiles fit in LDS
    // (experimental, slower on both bench workloads: VALU staging and f32 MFMA share the SIMD's ALUs, so the
    //  overlap it enforces buys nothing; kept for A/B runs) QRL_DECIM_TWO_TEAM=1
    const char* two = std::getenv("QRL_DECIM_TWO_TEAM");
    if (two && two[0] == '1' && q.in && q.n >= 2 && NLD <= 16 && mfma2_lds(q.nt, q.D, NA) <= 160 * 1024) {
        launch_k2<NA, (NLD <= 16 ? NLD : 16)>(q, grid, mfma2_lds(q.nt, q.D, NA), s);
        return;
    }
    // FAST: the tile comes from the caller's buffer through register-prefetched 16-byte loads
    if (q.in && q.n >= 2 && q.n < (1u << 28)) launch_k<NA, NLD, true>(q, grid, lds, s);
    else launch_k<NA, 1, false>(q, grid, lds, s);
}

void launch_decim_mfma(const DecimParams& p, int batch, hipStream_t s)
{
    if (p.m_count == 0) return;
    const int NA = decim_mfma_na(p.nt, p.D);
    const uint32_t T = 16 * NA;
    const uint32_t tiles = (uint32_t)((p.m0 + p.m_count + T - 1) / T - p.m0 / T);
    DecimParams q = p;
    q.tiles = tiles;
    // consecutive tiles per workgroup: enough to amortise the un-overlapped first load, few enough to keep >= ~4k workgroups
    const uint64_t total = (uint64_t)tiles * (uint64_t)batch;
    uint32_t tpw = (uint32_t)std::min<uint64_t>(kTpwMax, std::max<uint64_t>(1, total / 4096));
    q.tpw = tpw;
    q.nhi = mfma_nhi(p.nt, p.D, NA, tpw);
    q.magic_blk = (uint32_t)((0x100000000ull + 16u * (uint32_t)p.D - 1) / (16u * (uint32_t)p.D));
    q.magic_seg = (uint32_t)((0x100000000ull + 4u * (uint32_t)p.D - 1) / (4u * (uint32_t)p.D));
    const uint32_t chunks = (tiles + tpw - 1) / tpw;
    q.nchunks = chunks;
    { const char* e = std::getenv("QRL_DBG"); q.dbg = e ? std::atoi(e) : 0; }
    // never pad a short grid.x to 8: the padding blocks would leave whole XCDs idle
    dim3 grid(chunks >= 64 ? (chunks + 7) / 8 * 8 : chunks, batch);
    const size_t lds = mfma_lds(p.nt, p.D, NA, tpw);
    const long long pairs = (mfma_jtot(p.nt, p.D, NA) + 2) / 2;
    const int nld = (int)((pairs + 255) / 256);
    const bool fast = q.in && q.n >= 2 && q.n < (1u << 28);
    const char* w8 = std::getenv("QRL_DECIM_W8");   // 8 waves per workgroup (4 per SIMD with two workgroups per CU)
    const bool wide = w8 && w8[0] == '1' && fast && nld <= 16;
    // More than 16 loads per thread (front ends beyond ~40:1): a 36-load variant would need 144 prefetch registers, more than
    // the accumulator half of the register file holds, and hipcc then spills registers whose asm-issued loads are still in
    // flight.  Those tiles run with 512 threads (<= 16 loads per thread); anything larger takes the slow per-sample staging.
    const int nld512 = (int)((pairs + 511) / 512);
    const bool big = fast && nld > 16 && nld512 <= 16;
    q.nld = (wide || big) ? nld512 : nld;
    if (fast && nld > 16 && !big) {
        if (NA == 16) launch_k<16, 1, false>(q, grid, lds, s); else if (NA == 8) launch_k<8, 1, false>(q, grid, lds, s); else launch_k<4, 1, false>(q, grid, lds, s);
        return;
    }
    if (NA == 16) {
        if (wide) launch_k<16, 8, true, 4, 512>(q, grid, lds, s);
        else if (big) launch_k<16, 16, true, 2, 512>(q, grid, lds, s);
        else launch_one<16, 16>(q, grid, lds, s);
    } else if (NA == 4) {
        if (big) launch_k<4, 16, true, 2, 512>(q, grid, lds, s);
        else if (nld <= 8 && fast) launch_k<4, 8, true, 3>(q, grid, lds, s);
        else launch_one<4, 16>(q, grid, lds, s);
    } else {
        if (wide) { launch_k<8, 8, true, 4, 512>(q, grid, lds, s); return; }
        if (big) { launch_k<8, 16, true, 2, 512>(q, grid, lds, s); return; }
        const char* w3 = std::getenv("QRL_DECIM_WPE3");   // experiment: three smaller workgroups per CU
        if (w3 && w3[0] == '1' && nld <= 10 && fast) launch_k<8, 10, true, 3>(q, grid, lds, s);
        else launch_one<8, 16>(q, grid, lds, s);
    }
}

}  // namespace qrl
