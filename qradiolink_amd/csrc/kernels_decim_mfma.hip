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
#include <atomic>
#include <cstdlib>
#include "devmath.hpp"
#include "engine.hpp"

namespace qrl {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// phase profile of k_decim_mfma (developer aid, decim_mfma_prof_enable): shader-clock ticks summed over wave 0 of every workgroup
__device__ unsigned long long g_mf_prof[8];
static std::atomic<int> g_mf_prof_on{0};
void decim_mfma_prof_enable(int on) { g_mf_prof_on.store(on ? 32 : 0); }

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

constexpr int MF_U = 8;
typedef __attribute__((address_space(3))) const void* lds_cptr;

// One contiguous piece of the step loop (no pad jump inside): step i reads A = ap[-4 i] and B = bp[BS i].  Ping-pong over two
// register sets, compiler-scheduled (the compiler owns every lgkmcnt wait), sched_group_barrier pinned: one LDS read group after
// every MFMA.  (A hand-written asm variant of this loop with explicit waits was ~5 % faster but produced rare wrong outputs with
// two workgroups per CU that were never explained; it was removed rather than kept behind a switch.)
template <typename BT, int BS>
struct MfChunk {
    float a[MF_U]; BT b[MF_U];
    __device__ __forceinline__ void load(const float* ap, const BT* bp, int i)
    {
#pragma unroll
        for (int u = 0; u < MF_U; ++u) {
            // volatile LDS-address-space reads: hipcc must not fuse neighbours into ds_read2_b32 / ds_read2_b64, which the
            // LDS serves at half rate (MI355X_MICROARCH.md, LDS table).  The compiler still owns the waitcnt bookkeeping.
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
        }
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
__device__ __forceinline__ void mfma_piece_c(const float* __restrict__ ap, const BT* __restrict__ bp, int n, f32x4& acc0, f32x4& acc1)
{
    constexpr int U = MF_U;
    int i = 0;
    if (n >= U) {
        MfChunk<BT, BS> r0, r1;
        r0.load(ap, bp, 0);
        while (i + 2 * U <= n) {
            r1.load(ap, bp, i + U);
            mf_fma(r0, acc0, acc1);
            // the wave issues in order and blocks at an MFMA while the pipe is busy (32 cycles per v_mfma_f32_16x16x4_f32, one
            // dependent chain already runs at that rate: tools/ubench/mfma_chain.hip); everything issued BETWEEN two MFMAs is
            // free, a block of operand reads behind 16 MFMAs is not.  Interleave: one LDS read group after every MFMA.
            {
                constexpr int NM = BS == 4 ? 2 * U : U, RD = 2 * U / NM;   // MFMAs per chunk; operand reads per MFMA
#pragma unroll
                for (int k = 0; k < NM; ++k) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, RD, 0); }
            }
            __builtin_amdgcn_sched_barrier(0);
            r0.load(ap, bp, i + 3 * U <= n ? i + 2 * U : 0);   // beyond the end: harmless re-read of chunk 0
            mf_fma(r1, acc0, acc1);
            {
                constexpr int NM = BS == 4 ? 2 * U : U, RD = 2 * U / NM;
#pragma unroll
                for (int k = 0; k < NM; ++k) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 1); __builtin_amdgcn_sched_group_barrier(0x100, RD, 1); }
            }
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
            mfma_piece_c<float2, 4>(ap - 4 * s, bp + 4 * s + 2 * seg, e - s, acc0, acc1);
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
            mfma_piece_c<float, 8>(ap - 4 * s, bp + 2 * (4 * s + 2 * seg), e - s, acc0, acc1);
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
// 16 output blocks per tile when two workgroups of that size fit the 160 KB of a CU, else 8
int decim_mfma_na(int nt, int D)
{
    if (mfma_lds(nt, D, 16, kTpwMax) <= 80 * 1024) return 16;    // two (or more) workgroups per CU with the efficient tile
    if (mfma_lds(nt, D, 8, kTpwMax) <= 160 * 1024) return 8;   // (4-block tiles with three workgroups per CU were measured slower)
    return 4;   // very long filters (100:1 front end, 4181 taps): only the 4-block tile fits the 160 KB of a CU
}
size_t decim_mfma_lds_bytes(int nt, int D) { return mfma_lds(nt, D, decim_mfma_na(nt, D), kTpwMax); }

template <int NA, int NLD, bool FAST, int WPE = 2, int NTH = 256>
static hipError_t launch_k(const DecimParams& q, dim3 grid, size_t lds, hipStream_t s)
{
    const auto kern = k_decim_mfma<NA, NLD, FAST, true, WPE, NTH>;
    const hipError_t e = dyn_lds_limit(reinterpret_cast<const void*>(kern), 160 * 1024);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, grid, dim3(NTH), lds + (NTH - 256) * 2 * sizeof(float2), s, q);
    return hipSuccess;
}
template <int NA, int NLD>
static hipError_t launch_one(const DecimParams& q, dim3 grid, size_t lds, hipStream_t s)
{
    // FAST: the tile comes from the caller's buffer through register-prefetched 16-byte loads
    if (q.in && q.n >= 2 && q.n < (1u << 28)) return launch_k<NA, NLD, true>(q, grid, lds, s);
    return launch_k<NA, 1, false>(q, grid, lds, s);
}

int launch_decim_mfma(const DecimParams& p, int batch, hipStream_t s)
{
    if (p.m_count == 0) return 0;
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
    q.dbg = g_mf_prof_on.load(std::memory_order_relaxed);
    // never pad a short grid.x to 8: the padding blocks would leave whole XCDs idle
    dim3 grid(chunks >= 64 ? (chunks + 7) / 8 * 8 : chunks, batch);
    const size_t lds = mfma_lds(p.nt, p.D, NA, tpw);
    const long long pairs = (mfma_jtot(p.nt, p.D, NA) + 2) / 2;
    const int nld = (int)((pairs + 255) / 256);
    const bool fast = q.in && q.n >= 2 && q.n < (1u << 28);
    // More than 16 loads per thread (front ends beyond ~40:1): a 36-load variant would need 144 prefetch registers, more than
    // the accumulator half of the register file holds, and hipcc then spills registers whose asm-issued loads are still in
    // flight.  Those tiles run with 512 threads (<= 16 loads per thread); anything larger takes the slow per-sample staging.
    const int nld512 = (int)((pairs + 511) / 512);
    const bool big = fast && nld > 16 && nld512 <= 16;
    q.nld = big ? nld512 : nld;
    hipError_t e;
    if (fast && nld > 16 && !big) {
        e = NA == 16 ? launch_k<16, 1, false>(q, grid, lds, s) : NA == 8 ? launch_k<8, 1, false>(q, grid, lds, s) : launch_k<4, 1, false>(q, grid, lds, s);
    } else if (NA == 16) {
        e = big ? launch_k<16, 16, true, 2, 512>(q, grid, lds, s) : launch_one<16, 16>(q, grid, lds, s);
    } else if (NA == 4) {
        e = big ? launch_k<4, 16, true, 2, 512>(q, grid, lds, s) : (nld <= 8 && fast) ? launch_k<4, 8, true, 3>(q, grid, lds, s) : launch_one<4, 16>(q, grid, lds, s);
    } else {
        e = big ? launch_k<8, 16, true, 2, 512>(q, grid, lds, s) : launch_one<8, 16>(q, grid, lds, s);
    }
    return e == hipSuccess ? 0 : -1;
}

}  // namespace qrl
