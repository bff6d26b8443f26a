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
        return P.hist[(size_t)b * P.hist_len + (P.hist_len - (uint32_t)d)];
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
template <int NLD>
__device__ __forceinline__ void tile_issue(const DecimParams& P, int b, const TileWin& w, int tid, f32x4 (&v)[NLD])
{
    // UNCONDITIONAL loads with the pair index clamped into the caller's buffer; lanes outside
    // [k_lo, k_hi) fetch some valid pair, tile_commit sends them to the dump slots.
    const float4* src = reinterpret_cast<const float4*>(P.in + (size_t)b * P.in_stride);
    const int64_t q0 = (w.i_base - w.a - (int64_t)P.n0) >> 1;      // pair index of k = 0 (may be negative)
    const int64_t qmax = (int64_t)(P.n >> 1) - 1;
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
        int64_t q = q0 + w.k_lo + tid + 256 * it;
        q = q < 0 ? 0 : (q > qmax ? qmax : q);
        const float4* p = src + q;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(v[it]) : "v"(p) : "memory");
    }
}
template <int NLD>
__device__ __forceinline__ void tile_wait(f32x4 (&v)[NLD])
{
#pragma unroll
    for (int it = 0; it < NLD; ++it) asm volatile("s_waitcnt vmcnt(0)" : "+a"(v[it]) : : "memory");
}

template <int NLD, bool FAST>
__device__ __forceinline__ void tile_commit(const DecimParams& P, int b, const TileWin& w, int tid, const f32x4 (&v)[NLD],
                                            float2* tile, int Jtot, int dump, const float2* t_hi, uint32_t kb0, const float2* t_lo)
{
    const uint32_t magic = P.magic_blk;   // ceil(2^32 / (16 D)): j / (16 D) == umulhi(j, magic) for j < 2^16
    // samples that cannot come from the aligned-pair path (history, stream edges): rare, one at a time
    for (int seg = 0; seg < 2; ++seg) {
        const int sb = seg ? w.s_hi : 0, se = seg ? Jtot : w.s_lo;
        for (int j = sb + tid; j < se; j += 256)
            tile[j + 2 * (int)__umulhi((uint32_t)j, magic)] = mf_fetch(P, b, w.i_base + j, t_hi, kb0, t_lo);
    }
    if (!FAST) return;
    // register-prefetched pairs: STRAIGHT-LINE code (no exec branches, so hipcc batches the LDS reads);
    // pairs past the end of the window are written to a per-thread dump slot behind the tile
    const bool rot = P.rot_enable != 0;
    const int j0 = 2 * (w.k_lo + tid) - w.a;
    const uint32_t krel0 = (uint32_t)((uint64_t)(w.i_base + j0) - P.rot_nbase - ((uint64_t)kb0 << 9));
    // krel advances by 512 per step: the fine-table factors of a thread never change
    const float2 lo0 = t_lo[krel0 & 511u], lo1 = t_lo[(krel0 + 1u) & 511u];
    const int npair = w.k_hi - w.k_lo - tid;   // this thread owns pairs it < ceil(npair / 256)
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
        const int j = j0 + 512 * it;
        const uint32_t krel = krel0 + 512u * it;
        float2 x0 = make_float2(v[it].x, v[it].y), x1 = make_float2(v[it].z, v[it].w);
        if (rot) {
            x0 = cmul_fma(x0, cmul_fma(t_hi[krel >> 9], lo0));
            x1 = cmul_fma(x1, cmul_fma(t_hi[(krel + 1u) >> 9], lo1));
        }
        const bool ok = 256 * it < npair;
        const int p0 = ok ? j + 2 * (int)__umulhi((uint32_t)j, magic) : dump + 2 * tid;
        const int p1 = ok ? j + 1 + 2 * (int)__umulhi((uint32_t)(j + 1), magic) : dump + 2 * tid + 1;
        tile[p0] = x0;
        tile[p1] = x1;
    }
}

// One contiguous piece of the step loop (no pad jump inside): step i reads A = ap[-4 i] and B = bp[BS i].
// Software pipelined by hand with two register sets: the LDS reads of chunk c+1 are issued before the
// MFMAs of chunk c; sched_barrier keeps hipcc from sinking the reads next to their uses.
constexpr int MF_U = 8;
template <typename BT, int BS>
struct MfChunk {
    float a[MF_U]; BT b[MF_U];
    __device__ __forceinline__ void load(const float* ap, const BT* bp, int i)
    {
#pragma unroll
        for (int u = 0; u < MF_U; ++u) { a[u] = ap[-4 * (i + u)]; b[u] = bp[BS * (i + u)]; }
    }
};
__device__ __forceinline__ void mf_fma(const MfChunk<float2, 4>& c, f32x4& acc0, f32x4& acc1)
{
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
__device__ __forceinline__ void mfma_piece(const float* __restrict__ ap, const BT* __restrict__ bp, int n, f32x4& acc0, f32x4& acc1)
{
    constexpr int U = MF_U;
    int i = 0;
    if (n >= U) {
        MfChunk<BT, BS> r0, r1;
        r0.load(ap, bp, 0);
        while (i + 2 * U <= n) {
            r1.load(ap, bp, i + U);
            __builtin_amdgcn_sched_barrier(0);
            mf_fma(r0, acc0, acc1);
            __builtin_amdgcn_sched_barrier(0);
            r0.load(ap, bp, i + 3 * U <= n ? i + 2 * U : 0);   // beyond the end: harmless re-read of chunk 0
            __builtin_amdgcn_sched_barrier(0);
            mf_fma(r1, acc0, acc1);
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

template <int NA, int NLD, bool FAST>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_decim_mfma(const DecimParams P_)
{
    const DecimParams& P = P_;
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int T = 16 * NA;
    const int D = P.D, S = P.S, Sq = S >> 2;
    const int blk = 16 * D;
    const int Pp = blk + 2;
    const int Jtot = (NA - 1) * blk + 4 * S;
    const int hpn = (15 * D + 4 * S + 4 + 3) & ~3;   // multiple of 4 floats: the tile behind it stays 16-byte aligned
    float2* t_lo = reinterpret_cast<float2*>(smem);            // 512
    float2* t_hi = t_lo + 512;                                 // P.nhi
    float* hp = reinterpret_cast<float*>(t_hi + P.nhi);        // zero-padded taps, hpn (multiple of 4) floats
    float2* tile = reinterpret_cast<float2*>(hp + hpn);
    float2* part = tile;                                       // 4 * T partial sums alias the head of the tile
    const int dump = Jtot + 2 * (Jtot / blk) + 4;              // 512 dump slots behind the tile (tile_commit)

    const int b = blockIdx.y;
    const uint32_t per = gridDim.x >> 3;
    const uint32_t cix = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);   // neighbouring chunks share an XCD/L2
    // workgroup cix walks tiles cix, cix + nchunks, cix + 2 nchunks, ...: the workgroups resident at any
    // moment sweep ONE contiguous region of the stream together (no HBM channel camping, halos shared in L2)
    const uint32_t nchunks = P.nchunks;
    if (cix >= nchunks || cix >= P.tiles) return;
    const uint32_t t0 = cix;
    const int tid = threadIdx.x;
    const uint64_t mt_first = (P.m0 / T) * (uint64_t)T;
    const uint64_t mt0 = mt_first + (uint64_t)t0 * T;

    for (int k = tid; k < hpn; k += 256) hp[k] = P.gtab[k];
    uint32_t kb0 = 0;
    if (P.rot_enable) {
        t_lo[tid] = P.rot_lo[tid];
        t_lo[tid + 256] = P.rot_lo[tid + 256];
    }

    const int g = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int kk = lane >> 4;
    const uint32_t magic_seg = P.magic_seg;   // ceil(2^32 / (4 D))

    f32x4 v[NLD];
    TileWin w = tile_window(P, mt0, Jtot, FAST);
    if constexpr (FAST) { if (!(P.dbg & 4)) tile_issue<NLD>(P, b, w, tid, v); }

    const bool prof = (P.dbg & 32) && tid == 0;
    unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = prof ? __builtin_readcyclecounter() : 0;
#define MF_STAMP(k) do { if (prof) { const unsigned long long tn_ = __builtin_readcyclecounter(); pc[k] += tn_ - tprev; tprev = tn_; } } while (0)
    for (uint32_t t = t0; t < P.tiles; t += nchunks) {
        const uint64_t mt = mt_first + (uint64_t)t * T;
        if (P.rot_enable) {   // coarse rotator table of this tile (previous tile's readers are past the last barrier)
            const int64_t first_new = w.i_base > (int64_t)P.n0 ? w.i_base : (int64_t)P.n0;
            kb0 = (uint32_t)(((uint64_t)first_new - P.rot_nbase) >> 9);
            if (tid < P.nhi) t_hi[tid] = sincos_turn(P.rot_acc + ((uint64_t)(kb0 + tid) << 9) * P.rot_inc);
        }
        __syncthreads();
        MF_STAMP(0);
        if constexpr (FAST) tile_wait<NLD>(v);
        if (!(P.dbg & 2)) tile_commit<NLD, FAST>(P, b, w, tid, v, tile, Jtot, dump, t_hi, kb0, t_lo);
        MF_STAMP(1);
        __syncthreads();
        MF_STAMP(2);
        if (t + nchunks < P.tiles) {   // next tile's loads fly during the MFMA phase
            w = tile_window(P, mt + (uint64_t)nchunks * T, Jtot, FAST);
            if constexpr (FAST) { if (!(P.dbg & 4)) tile_issue<NLD>(P, b, w, tid, v); }
        }

        MF_STAMP(3);
        // ---- FIR on the matrix pipe: wave g = quarter g of the u axis ----
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        int s = g * Sq;
        const int s_end = (P.dbg & 1) ? s : s + Sq;
        const int seg_len = 4 * D;   // steps between two pad jumps of the B operand
        if constexpr (NA == 16) {
            const int acol = lane & 15;
            const float2* bp = tile + ((P.dbg & 8) ? 0 : acol * Pp) + kk;
            const float* ap = hp + (((P.dbg & 16) ? 0 : acol * D) + 4 * S - kk);     // tap k = b D - u, stored at k + (4 S - nt + 1)
            while (s < s_end) {
                const int seg = (int)__umulhi((uint32_t)s, magic_seg);
                const int e = min(s_end, (seg + 1) * seg_len);
                mfma_piece<float2, 4>(ap - 4 * s, bp + 4 * s + 2 * seg, e - s, acc0, acc1);
                s = e;
            }
            MF_STAMP(4);
            __syncthreads();   // every wave is done reading the tile: its head becomes the partial-sum area
            MF_STAMP(5);
            float2* pp = part + g * T + 16 * acol + 4 * kk;
#pragma unroll
            for (int r = 0; r < 4; ++r) pp[r] = make_float2(acc0[r], acc1[r]);
        } else {
            const int n = lane & 15, acol = n >> 1, c = n & 1;
            const float* bp = reinterpret_cast<const float*>(tile) + 2 * (acol * Pp + kk) + c;
            const int brow = lane & 15;
            const float* ap = hp + (brow * D + 4 * S - kk);
            while (s < s_end) {
                const int seg = (int)__umulhi((uint32_t)s, magic_seg);
                const int e = min(s_end, (seg + 1) * seg_len);
                mfma_piece<float, 8>(ap - 4 * s, bp + 2 * (4 * s + 2 * seg), e - s, acc0, acc1);
                s = e;
            }
            __syncthreads();
            float* pf = reinterpret_cast<float*>(part + g * T + 16 * acol + 4 * kk) + c;
#pragma unroll
            for (int r = 0; r < 4; ++r) pf[2 * r] = acc0[r];
        }
        __syncthreads();
        if (tid < T) {
            const uint64_t m = mt + tid;
            if (m >= P.m0 && m < P.m0 + P.m_count) {
                const float2 r0 = part[tid], r1 = part[T + tid], r2 = part[2 * T + tid], r3 = part[3 * T + tid];
                float2 y;
                y.x = (r0.x + r1.x) + (r2.x + r3.x);
                y.y = (r0.y + r1.y) + (r2.y + r3.y);
                P.out.p[(size_t)b * (P.out.mask + 1u) + ((uint32_t)m & P.out.mask)] = y;
            }
        }
        __syncthreads();   // partial sums consumed before the next tile overwrites them
        MF_STAMP(6);
    }
    if (prof) {
#pragma unroll
        for (int k = 0; k < 8; ++k) atomicAdd(&g_mf_prof[k], pc[k]);
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
    const long long npos = Jtot + 2 * (Jtot / (16LL * D)) + 4 + 512;   // + dump slots
    return (size_t)(512 + mfma_nhi(nt, D, NA, tpw) + npos) * sizeof(float2) + (size_t)decim_mfma_hpn(nt, D) * sizeof(float);
}
// rule shared with oracle/orc_blocks.c orc_decim_uses_m16
bool decim_uses_mfma(int nt, int D)
{
    if (D < 8) return false;
    const long long samples = 7LL * 16 * D + 4LL * decim_mfma_steps(nt, D);
    return (samples + 2 * (samples / (16LL * D)) + 64) * 8 <= 150 * 1024;
}
constexpr int kTpwMax = 16;
int decim_mfma_na(int nt, int D) { return mfma_lds(nt, D, 16, kTpwMax) <= 80 * 1024 ? 16 : 8; }
size_t decim_mfma_lds_bytes(int nt, int D) { return mfma_lds(nt, D, decim_mfma_na(nt, D), kTpwMax); }

template <int NA, int NLD, bool FAST>
static void launch_k(const DecimParams& q, dim3 grid, size_t lds, hipStream_t s)
{
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_decim_mfma<NA, NLD, FAST>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL((k_decim_mfma<NA, NLD, FAST>), grid, dim3(256), lds, s, q);
}
template <int NA, int NLD>
static void launch_one(const DecimParams& q, dim3 grid, size_t lds, hipStream_t s)
{
    // FAST: the tile comes from the caller's buffer through register-prefetched 16-byte loads
    if (q.in && q.n >= 2) launch_k<NA, NLD, true>(q, grid, lds, s);
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
    dim3 grid((chunks + 7) / 8 * 8, batch);
    const size_t lds = mfma_lds(p.nt, p.D, NA, tpw);
    const long long pairs = (mfma_jtot(p.nt, p.D, NA) + 2) / 2;
    const int nld = (int)((pairs + 255) / 256);
    if (NA == 16) {
        if (nld <= 16) launch_one<16, 16>(q, grid, lds, s); else launch_one<16, 36>(q, grid, lds, s);
    } else {
        if (nld <= 16) launch_one<8, 16>(q, grid, lds, s); else launch_one<8, 36>(q, grid, lds, s);
    }
}

}  // namespace qrl
