// kernels_chan.hip — multi-carrier front end of the MMDVM path (reference src/gr/gr_demod_mmdvm_multi2.cpp:98-101):
//   stream_to_streams(M) + pfb_channelizer_ccf(M, taps, 1.0)  ->  M channels at fs / M, channel c centred at +c fs/M.
//  k_pfb_chan : workgroup = TI output instants of one wideband stream.  Phase 1: the M polyphase branch FIRs
//     v_p[n] = sum_k h[p + M k] x[M n - p - M k] (one fmaf chain per branch, k ascending) out of an LDS copy of the
//     input tile.  Phase 2: the M-point DFT y_c = sum_p v_p W[(p c) mod M], written as the direct sum the oracle
//     defines (upstream runs FFTW; every FFT factorisation rounds differently).  Only the channels
//     [c_first, c_first + c_count) are produced: a rank of a channel-sharded job computes just its own bins.
//     The input is read from HBM exactly once; the tile halo (nt - 1 samples) comes from L2 / the history buffer.
//  k_f2s      : multiply_const_ff(level) + float_to_short(1, 32767) (gr_demod_mmdvm_multi2.cpp:84,92).
#include <algorithm>
#include <mutex>
#include <cstdlib>
#include "devmath.hpp"
#include "engine.hpp"

namespace qrl {

constexpr int CH_TI = 64;   // output instants per workgroup

// where channel cc (relative to c_first) of stream b, output instant m goes: an engine ring row, or (out_pitch > 0) a linear caller
// buffer; row_cpd > 0 groups the rows by destination rank for an all-to-all (engine.hpp ChanParams)
__device__ __forceinline__ float2* chan_out_addr(const ChanParams& P, int b, int nbatch, int cc, uint64_t m)
{
    const size_t rowi = P.row_cpd ? ((size_t)(cc / (int)P.row_cpd) * nbatch + b) * P.row_cpd + (cc % (int)P.row_cpd) : (size_t)b * P.c_count + cc;
    const size_t col = P.out_pitch ? (size_t)(m - P.m0) : (size_t)((uint32_t)m & P.out.mask);
    return P.out.p + rowi * (P.out_pitch ? P.out_pitch : (size_t)P.out.mask + 1u) + col;
}

typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <int STEPS>   // STEPS = M / 4 when M is a multiple of 16 (MFMA DFT), 0 = any M (VALU DFT)
__global__ __launch_bounds__(256) void k_pfb_chan(const ChanParams P)
{
    extern __shared__ __align__(16) unsigned char ch_smem[];
    const int M = P.M, J = P.J;
    float2* xs = reinterpret_cast<float2*>(ch_smem);              // (TI + J) * M input samples, xs[i] = x[first + i]
    float2* vs = xs + (CH_TI + J) * M;                            // [TI][M + 1] branch outputs
    float* taps = reinterpret_cast<float*>(vs + CH_TI * (M + 1)); // J * M (zero padded)
    float2* W = reinterpret_cast<float2*>(taps + J * M);          // M twiddles
    const int b = blockIdx.y, tid = threadIdx.x;
    const uint64_t m_t = P.m0 + (uint64_t)blockIdx.x * CH_TI;     // first output instant of this tile (absolute)
    const int64_t first = (int64_t)m_t * M - (int64_t)(J * M - 1);// oldest sample any branch of the tile reads
    const int nsamp = (CH_TI + J) * M;
    for (int i = tid; i < J * M; i += 256) taps[i] = P.taps[i];
    for (int i = tid; i < M; i += 256) W[i] = P.twiddle[i];
    for (int i = tid; i < nsamp; i += 256) {
        const int64_t a = first + i;
        float2 x = make_float2(0.f, 0.f);
        if (a >= 0 && (uint64_t)a < P.n0 + P.n) {
            if ((uint64_t)a >= P.n0) x = P.in[(size_t)b * P.in_stride + (size_t)((uint64_t)a - P.n0)];
            else {
                const uint64_t d = P.n0 - (uint64_t)a;
                if (d <= P.hist_len) x = P.hist[(size_t)b * P.hist_len + (P.hist_len - (uint32_t)d)];
            }
        }
        xs[i] = x;
    }
    __syncthreads();
    // phase 1: branch (i, p) reads x[M (m_t + i) - p - M k] = xs[(J * M - 1) + M i - p - M k]
    for (int w = tid; w < CH_TI * M; w += 256) {
        const int i = w / M, p = w - i * M;
        const float2* xp = xs + (J * M - 1) + M * i - p;
        float ar = 0.f, ai = 0.f;
        for (int k = 0; k < J; ++k) {
            const float h = taps[p + M * k];
            const float2 x = xp[-M * k];
            ar = fmaf(h, x.x, ar);
            ai = fmaf(h, x.y, ai);
        }
        vs[i * (M + 1) + p] = make_float2(ar, ai);
    }
    __syncthreads();
    // phase 2: DFT bins of the owned channels.  Contract (oracle orc_pfb_channelizer): four real fmaf chains over the branches,
    // p ascending -- sa = sum W.re v.re, sb = sum W.im v.im, sc = sum W.im v.re, sd = sum W.re v.im -- y = (sa - sb, sc + sd).
    const uint32_t count = P.m_count;
    if constexpr (STEPS > 0) {
        // M = 4 STEPS is a multiple of 16: the DFT is a [bins x branches] x [branches x instants] product on the f32 matrix
        // pipe.  One MFMA block = 16 bins x 16 instants; A operand = twiddles W[(bin p) mod M] (rebuilt per bin block, held in
        // registers), B operand = the branch outputs of phase 1 (one ds_read_b64 feeds the .re and the .im chains); four
        // independent accumulators per block.  f32 MFMA accumulates in k order exactly like the fmaf chains of the contract.
        const int lane = tid & 63, wv = tid >> 6;
        const int qb0 = P.c_first >> 4, qb1 = (P.c_first + P.c_count - 1) >> 4;
        const int nitems = (qb1 - qb0 + 1) * (CH_TI / 16);
        for (int it = wv; it < nitems; it += 4) {
            const int qb = qb0 + it / (CH_TI / 16), ib = it % (CH_TI / 16);
            const int bin_a = 16 * qb + (lane & 15), k4 = lane >> 4;
            float are[STEPS], aim[STEPS];
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                const float2 wv2 = W[(bin_a * (4 * s + k4)) % M];
                are[s] = wv2.x; aim[s] = wv2.y;
            }
            const float2* vb = vs + (16 * ib + (lane & 15)) * (M + 1) + k4;
            f32x4_t sa = {0.f, 0.f, 0.f, 0.f}, sb = sa, sc = sa, sd = sa;
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                const float2 v = vb[4 * s];
                sa = __builtin_amdgcn_mfma_f32_16x16x4f32(are[s], v.x, sa, 0, 0, 0);
                sb = __builtin_amdgcn_mfma_f32_16x16x4f32(aim[s], v.y, sb, 0, 0, 0);
                sc = __builtin_amdgcn_mfma_f32_16x16x4f32(aim[s], v.x, sc, 0, 0, 0);
                sd = __builtin_amdgcn_mfma_f32_16x16x4f32(are[s], v.y, sd, 0, 0, 0);
            }
            const int i = 16 * ib + (lane & 15);
            if (blockIdx.x * (uint32_t)CH_TI + i < count) {
                const uint64_t m = m_t + i;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int bin = 16 * qb + 4 * k4 + r, cc = bin - P.c_first;
                    if (cc >= 0 && cc < P.c_count)
                        *chan_out_addr(P, b, gridDim.y, cc, m) = make_float2(sa[r] - sb[r], sc[r] + sd[r]);
                }
            }
        }
    } else {
        for (int w = tid; w < CH_TI * P.c_count; w += 256) {
            const int cc = w / CH_TI, i = w - cc * CH_TI;             // instant fastest: coalesced ring writes
            const int c = P.c_first + cc;
            if (blockIdx.x * (uint32_t)CH_TI + i >= count) continue;
            const float2* v = vs + i * (M + 1);
            float sa = 0.f, sb = 0.f, sc = 0.f, sd = 0.f;
            int q = 0;                                                // (p * c) mod M
            for (int p = 0; p < M; ++p) {
                const float2 wv = W[q];
                sa = fmaf(wv.x, v[p].x, sa);
                sb = fmaf(wv.y, v[p].y, sb);
                sc = fmaf(wv.y, v[p].x, sc);
                sd = fmaf(wv.x, v[p].y, sd);
                q += c; if (q >= M) q -= M;
            }
            const uint64_t m = m_t + i;
            *chan_out_addr(P, b, gridDim.y, cc, m) = make_float2(sa - sb, sc + sd);
        }
    }
}

// ---- k_pfb_stream64: the production channelizer of BASELINE config 4 since round 4 -- same contract as k_pfb_chan ---------------
// Round 3's k_pfb_chan64 cut a stream into 32-instant tiles, one workgroup each: every tile re-read a 34-instant halo (PMC: 2.07 x the
// input bytes), re-loaded the 35 taps of its branch and the twiddles, and spent as much matrix-pipe time on bins 33..63 as on their
// mirror images.  Here a workgroup is PERSISTENT: it walks one segment of one wideband stream (grid = segments x streams, sized to the
// chip: 3 workgroups per CU) in tiles of 16 output instants and
//   * keeps the input in an LDS RING of 84 blocks of 64 samples (42 KiB): per tile only the 16 NEW blocks are fetched, with LDS-DMA
//     (global_load_lds_dwordx4, eight 1 KiB pieces issued TWO tiles ahead: 51 live blocks + 32 in flight fit the ring, nothing
//     aliases; one tile in flight per workgroup left every tile waiting for the memory latency); the input is read from HBM once
//     (+ 35 blocks per segment);
//   * keeps the taps of the lane's branch (35 registers) and the wave's DFT operand (32 registers) for its whole life;
//   * phase 1 (VALU): lane = branch p, wave w = instants 4 w .. 4 w + 3: v_p[m] = sum_k h[p + 64 k] x[64 (m - k) - p], one packed-fma
//     chain per output (k ascending), every LDS sample feeding up to four chains; the block part of an address is wave uniform
//     (scalar ALU), one v_add per read;
//   * phase 2 (matrix pipe): wave = (16 bins, 8 instants).  The B operand carries v.re of the 8 instants in columns 0..7 and v.im in
//     columns 8..15, so TWO accumulators give all four chains of the contract: X = Wre * [vre | vim] = [sa | sd], Y = Wim * [vre | vim]
//     = [sc | sb]; a DPP row rotation by 8 brings the partner column: P = X - rot(Y), Q = Y + rot(X) are (sa - sb, sc + sd) = y[bin]
//     in the low columns and (sd - sc, sa + sb) in the high ones -- which is y[64 - bin] with re / im swapped, bit for bit: the
//     twiddle table is exactly conjugate symmetric (oracle orc_chan_twiddles), so the chains of bin 64 - c are those of bin c with
//     sb, sc negated.  Bins 1..31 and 33..63 therefore cost 32 matrix instructions per 16 x 8 outputs instead of 128; bin 32
//     (W = +-1, 0) is an alternating add chain on the VALU of one wave per tile.
// Ring position of block beta = (beta - (m_lo - 64)) mod RB: tile t owns positions (64 + 16 t) mod RB ...; lane p >= 1 reads block
// m - k - 1 at offset 64 - p, lane 0 block m - k at offset 0 = ONE SAMPLE behind the end of block m - k - 1: positions RB, RB + 1 mirror
// positions 0, 1 (the piece that lands there is issued twice), so that lane 0 needs no wrap of its own.
// Order inside a tile: issue the pieces of tile t + 2 -> FIRs -> barrier -> bin 32 and matrix phase -> s_waitcnt vmcnt(pieces just
// issued): everything OLDER has completed, i.e. tile t + 1's pieces and the previous tile's stores -> this tile's stores -> barrier:
// a wave never waits for the acknowledgement of stores it has just issued, and the count does not depend on how many stores ran.
#ifndef QRL_S64_AHEAD
#define QRL_S64_AHEAD 2          // tiles of input in flight per workgroup (1 or 2)
#endif
constexpr int S64_T = 16, S64_VP = 66;
constexpr int S64_AHEAD = QRL_S64_AHEAD;
constexpr int S64_RB = S64_AHEAD == 1 ? 80 : 84;                 // ring blocks: >= 51 live + 16 AHEAD in flight, a multiple of 4
typedef float v2f_ch __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) v2f_ch* s64_lds_v2;
__device__ __forceinline__ void s64_glds16(const void* gsrc, uint32_t lds_dst)
{
    // one LDS-DMA piece: 64 lanes x 16 B from per-lane global addresses to LDS[lds_dst + 16 lane]; M0 saved / restored in the statement
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void s64_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
#ifdef QRL_S64_PROF
// developer build (tools/chan_variants.sh name -DQRL_S64_PROF): shader-clock ticks per phase of the tile loop, summed over every wave
__device__ unsigned long long g_s64_prof[8];
#define S64_STAMP(k) do { const unsigned long long tn_ = __builtin_readcyclecounter(); pc[k] += tn_ - tprev; tprev = tn_; } while (0)
#else
#define S64_STAMP(k) do { } while (0)
#endif
__device__ __forceinline__ int s64_wrap(int e) { return e >= S64_RB ? e - S64_RB : e; }
template <int J>
__global__ __launch_bounds__(256, 3) void k_pfb_stream64(const ChanParams P, uint32_t seg_len)
{
    constexpr int M = 64;
    extern __shared__ __align__(16) unsigned char ch_smem[];
    float2* xs = reinterpret_cast<float2*>(ch_smem);              // ring: RB blocks x 64 samples + 2 mirror blocks
    float2* vs = xs + (S64_RB + 2) * 64;                          // [16 instants][VP] branch outputs
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint64_t m_end = P.m0 + P.m_count;
    const uint64_t m_lo = P.m0 + (uint64_t)blockIdx.x * seg_len;
    if (m_lo >= m_end) return;
    const uint64_t m_hi = m_lo + seg_len < m_end ? m_lo + seg_len : m_end;
    const int ntiles = (int)((m_hi - m_lo + S64_T - 1) / S64_T);
    const uint64_t nb_end = (P.n0 + P.n) >> 6;                    // blocks [n0 / 64, nb_end) lie in the caller's buffer
    const float2* row = P.in + (size_t)b * P.in_stride;
    const uint32_t xs_base = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)xs;
    if (xs_base != 0) __builtin_trap();                          // the kernel has no static LDS: the dynamic segment (= the ring) starts at 0
    // taps of this lane's branch and the wave's DFT operand (rows = bins 16 bb + (lane & 15), k = 4 s + (lane >> 4)): registers, once
    // (as PAIRS (h[2 i], h[2 i + 1]): the packed fma broadcasts one half through op_sel; written as {h, h} vectors the compiler keeps
    //  every tap twice, 70 registers)
    v2f_ch hp[(J + 1) / 2];
#pragma unroll
    for (int k = 0; k < J; k += 2) hp[k / 2] = v2f_ch{P.taps[lane + M * k], k + 1 < J ? P.taps[lane + M * (k + 1)] : 0.f};
    const int bb = wv & 1, oct = wv >> 1, n16 = lane & 15, k4 = lane >> 4;
    float are[16], aim[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const float2 w2 = P.twiddle[((16 * bb + n16) * (4 * s + k4)) & 63];
        are[s] = w2.x; aim[s] = w2.y;
    }
    // the 16 blocks of the tile that starts at instant a0 into ring positions r0 .. (mod RB): LDS-DMA when they lie inside the caller's
    // buffer (returns the number of pieces THIS wave issued), checked element loads otherwise (ragged end of a call; returns 0)
    auto fetch_tile = [&](uint64_t a0, int r0) -> int {
        if (a0 + 16 <= nb_end) {
            const unsigned char* g = reinterpret_cast<const unsigned char*>(row + (a0 * 64 - P.n0)) + lane * 16;
            int nd = 0;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = wv + 4 * jj, pos = s64_wrap(r0 + 2 * j);            // even: a piece never straddles the ring's end
                s64_glds16(g + j * 1024, (uint32_t)pos * 512u);
                ++nd;
                if (pos == 0) { s64_glds16(g + j * 1024, (uint32_t)S64_RB * 512u); ++nd; }   // mirror of positions 0, 1
            }
            return nd;
        }
        for (int i = tid; i < 16 * 64; i += 256) {
            const uint64_t sa_ = a0 * 64 + (uint64_t)i;
            const float2 x = sa_ < P.n0 + P.n ? row[(size_t)(sa_ - P.n0)] : make_float2(0.f, 0.f);
            const int pos = s64_wrap(r0 + (i >> 6));
            xs[pos * 64 + (i & 63)] = x;
            if (pos < 2) xs[(S64_RB + pos) * 64 + (i & 63)] = x;
        }
        return 0;
    };
    // prologue: halo blocks m_lo - 35 .. m_lo - 1 and the first tile's 16 blocks (ring positions 29 .. 79), checked element loads
    {
        const int64_t beta0 = (int64_t)m_lo - 64;
        for (int i = tid; i < 51 * 64; i += 256) {
            const int rel = 29 + (i >> 6);
            const int64_t a = (beta0 + rel) * 64 + (i & 63);
            float2 x = make_float2(0.f, 0.f);
            if (a >= 0 && (uint64_t)a < P.n0 + P.n) {
                if ((uint64_t)a >= P.n0) x = row[(size_t)((uint64_t)a - P.n0)];
                else {
                    const uint64_t d = P.n0 - (uint64_t)a;
                    if (d <= P.hist_len) x = P.hist[(size_t)b * P.hist_len + (P.hist_len - (uint32_t)d)];
                }
            }
            xs[rel * 64 + (i & 63)] = x;
        }
    }
    if (S64_AHEAD == 2 && ntiles > 1) (void)fetch_tile(m_lo + 16, s64_wrap(64 + 16));   // tile 1 (waited for at the end of tile 0)
    const uint32_t vlane = lane == 0 ? 512u : (uint32_t)(64 - lane) * 8u;      // byte offset relative to block (m - k - 1)
    // output rows of this lane's four matrix results: low columns (n16 < 8) -> bin 16 bb + 4 k4 + r, high columns -> its mirror image 64 - bin
    // (chan_out_addr's row arithmetic, once per workgroup: per tile only the column changes)
    const bool lo = n16 < 8;
    const int col = n16 & 7;
    const float* vsf = reinterpret_cast<const float*>(vs);
    const uint32_t bofs = (uint32_t)((8 * oct + col) * S64_VP + k4) * 2u + (uint32_t)(n16 >> 3);
    const size_t opitch = P.out_pitch ? P.out_pitch : (size_t)P.out.mask + 1u;
    auto out_row = [&](int cc) -> float2* {
        const size_t rowi = P.row_cpd ? ((size_t)(cc / (int)P.row_cpd) * gridDim.y + b) * P.row_cpd + (cc % (int)P.row_cpd) : (size_t)b * P.c_count + cc;
        return P.out.p + rowi * opitch;
    };
    float2* orow[4];
    uint32_t ovalid = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int bin = 16 * bb + 4 * k4 + r, cc = (lo ? bin : 64 - bin) - P.c_first;
        const bool ok = (lo || bin != 0) && cc >= 0 && cc < P.c_count;
        orow[r] = out_row(ok ? cc : 0);
        ovalid |= ok ? 1u << r : 0u;
    }
    const bool nyq_on = P.c_first <= 32 && 32 < P.c_first + P.c_count;
    float* nyq_row = reinterpret_cast<float*>(out_row(nyq_on ? 32 - P.c_first : 0)) + (lane & 1);
#ifdef QRL_S64_PROF
    unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_readcyclecounter();
#endif
    __syncthreads();
    int rb = 64;                                                  // ring position of the tile's first block: (64 + 16 t) mod RB
    for (int t = 0; t < ntiles; ++t) {
        const uint64_t a = m_lo + (uint64_t)t * S64_T;            // first instant of the tile
        // the tile AHEAD: its positions hold blocks a + 16 AHEAD - RB ..: older than the oldest live block a - 35
        int nd = 0;
        if (t + S64_AHEAD < ntiles) nd = fetch_tile(a + 16 * S64_AHEAD, s64_wrap(s64_wrap(rb + 16 * S64_AHEAD)));
        S64_STAMP(0);
        // ---- phase 1: branch FIRs of instants a + 4 wv + r
        {
            v2f_ch acc[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = v2f_ch{0.f, 0.f};
            const int e0 = rb + 4 * wv + (S64_RB - 1);            // (block m - k - 1 of r = k; + RB keeps the sum positive)
            // the J + 3 samples the four windows share, read in groups of NBATCH that are issued one group ahead of their use (left to
            // itself the compiler keeps two reads in flight and the phase waits for the LDS latency 19 times: 3700 of 9400 cycles per tile)
            constexpr int NX = J + 3, NBATCH = 10, NG = (NX + NBATCH - 1) / NBATCH;
            v2f_ch xv[NX];                                        // xv[i]: d = 3 - i
            auto load_group = [&](int g) {
#pragma unroll
                for (int i = g * NBATCH; i < (g + 1) * NBATCH && i < NX; ++i) {
                    const int e = s64_wrap(s64_wrap(e0 + 3 - i)); // wave uniform: scalar ALU
                    xv[i] = *(s64_lds_v2)(uintptr_t)((uint32_t)e * 512u + vlane);
                }
            };
            load_group(0);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) load_group(g + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = g * NBATCH; i < (g + 1) * NBATCH && i < NX; ++i) {
                    const int d = 3 - i;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int k = r - d;
                        if (k >= 0 && k < J) {                    // acc[r] = fma((h[k], h[k]), x, acc[r]): one rounding per component, as fmaf
                            if (k & 1) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc[r]) : "v"(hp[k / 2]), "v"(xv[i]));
                            else       asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[r]) : "v"(hp[k / 2]), "v"(xv[i]));
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) vs[(4 * wv + r) * S64_VP + lane] = make_float2(acc[r].x, acc[r].y);
        }
        S64_STAMP(1);
        __syncthreads();
        S64_STAMP(2);
        // ---- bin 32: W[(32 p) & 63] = (+1, 0), (-1, 0): sa, sd are alternating add chains, sb = sc = +0 (one wave per tile)
        float ynyq = 0.f;
        const bool nyq_here = wv == (t & 3) && nyq_on;
        if (nyq_here && lane < 32) {
            const float* vp = vsf + (lane >> 1) * (2 * S64_VP) + (lane & 1);
            float sgn = 0.f;
            // two batches of 8 reads in flight (software pipelined by hand; fully unrolled without the fences the compiler hoists all 64
            // reads into 64 registers, rolled it waits for every batch: 8 x LDS latency on the workgroup's critical path)
            float va[8], vb[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) va[i] = vp[2 * i];
#pragma unroll
            for (int p = 0; p < M; p += 16) {
#pragma unroll
                for (int i = 0; i < 8; ++i) vb[i] = vp[2 * (p + 8 + i)];
                asm volatile("" ::: "memory");
#pragma unroll
                for (int i = 0; i < 8; i += 2) { sgn = fmaf(1.0f, va[i], sgn); sgn = fmaf(-1.0f, va[i + 1], sgn); }
                if (p + 16 < M) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) va[i] = vp[2 * (p + 16 + i)];
                }
                asm volatile("" ::: "memory");
#pragma unroll
                for (int i = 0; i < 8; i += 2) { sgn = fmaf(1.0f, vb[i], sgn); sgn = fmaf(-1.0f, vb[i + 1], sgn); }
            }
            const float zero = 0.0f;
            ynyq = (lane & 1) ? zero + sgn : sgn - zero;          // (sa - sb, sc + sd)
        }
        // ---- phase 2: bins 16 bb .. 16 bb + 15 (and their mirror images) of instants a + 8 oct .. + 7
        f32x4_t X = {0.f, 0.f, 0.f, 0.f}, Y = X;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float v = vsf[bofs + 8 * s];
            X = __builtin_amdgcn_mfma_f32_16x16x4f32(are[s], v, X, 0, 0, 0);
            Y = __builtin_amdgcn_mfma_f32_16x16x4f32(aim[s], v, Y, 0, 0, 0);
        }
        S64_STAMP(3);
        // everything older than the pieces issued at the top of this tile has completed: the next tile's pieces (this wave's) have landed,
        // and the stores waited for are the PREVIOUS tile's -- a wave never waits for the acknowledgement of stores it has just issued
        if (S64_AHEAD == 1 || nd == 0) s64_wait_vm<0>();
        else if (nd == 2) s64_wait_vm<2>();
        else if (nd == 3) s64_wait_vm<3>();
        else s64_wait_vm<4>();
        S64_STAMP(4);
        {
            const uint64_t m = a + 8 * oct + col;
            const size_t ocol = P.out_pitch ? (size_t)(m - P.m0) : (size_t)((uint32_t)m & P.out.mask);
            const uint32_t okm = m < m_hi ? ovalid : 0u;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // (element copies first: __builtin_bit_cast applied to a vector-element expression reads element 0 with this compiler)
                const float xe = X[r], ye = Y[r];
                const float Xr = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(xe), 0x128, 0xf, 0xf, false));   // row_ror:8
                const float Yr = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ye), 0x128, 0xf, 0xf, false));
                const float Pv = xe - Yr, Qv = ye + Xr;
                if (okm >> r & 1u) orow[r][ocol] = lo ? make_float2(Pv, Qv) : make_float2(Qv, Pv);
            }
            if (nyq_here && lane < 32) {
                const uint64_t mn = a + (lane >> 1);
                if (mn < m_hi) nyq_row[2 * (P.out_pitch ? (size_t)(mn - P.m0) : (size_t)((uint32_t)mn & P.out.mask))] = ynyq;
            }
        }
        S64_STAMP(5);
        __syncthreads();                                          // everybody's pieces have landed; vs is free again
        S64_STAMP(6);
        rb = s64_wrap(rb + 16);
    }
#ifdef QRL_S64_PROF
    if (lane == 0) { for (int k = 0; k < 7; ++k) atomicAdd(&g_s64_prof[k], pc[k]); atomicAdd(&g_s64_prof[7], (unsigned long long)ntiles); }
#endif
}
#ifdef QRL_S64_PROF
extern "C" void qrl_s64_prof_read(unsigned long long* out8)
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_s64_prof), 8 * sizeof(unsigned long long));
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_s64_prof), z, sizeof z);
}
#endif
size_t stream64_lds_bytes() { return (size_t)((S64_RB + 2) * 64 + S64_T * S64_VP) * sizeof(float2); }

// caller buffer [rows][pitch] -> engine ring rows at absolute items [q0, q0 + count): how the per-channel-only handle (form 3) takes
// the channel samples an all-to-all delivered
__global__ __launch_bounds__(256) void k_ring_load(const float2* in, size_t pitch, RingC out, uint64_t q0, uint32_t count)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= count) return;
    const int r = blockIdx.y;
    out.p[(size_t)r * (out.mask + 1u) + ((uint32_t)(q0 + t) & out.mask)] = in[(size_t)r * pitch + t];
}
// engine ring rows, absolute items [q0, q0 + count) -> caller buffer [rows][cap] (+ counts[row]): the scope tap's mailbox copy
__global__ __launch_bounds__(256) void k_ring_store(RingC in, uint64_t q0, uint32_t count, float2* out, size_t cap, uint32_t* counts)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    const int r = blockIdx.y;
    if (t == 0 && counts) counts[r] = count < cap ? count : (uint32_t)cap;
    if (t >= count || t >= cap) return;
    out[(size_t)r * cap + t] = in.p[(size_t)r * (in.mask + 1u) + ((uint32_t)(q0 + t) & in.mask)];
}
void launch_ring_store(RingC in, uint64_t q0, uint32_t count, float2* out, size_t cap, uint32_t* counts, int rows, hipStream_t s)
{
    hipLaunchKernelGGL(k_ring_store, dim3(count ? (count + 255) / 256 : 1, rows), dim3(256), 0, s, in, q0, count, out, cap, counts);
}
void launch_ring_load(const float2* in, size_t pitch, RingC out, uint64_t q0, uint32_t count, int rows, hipStream_t s)
{
    if (!count) return;
    hipLaunchKernelGGL(k_ring_load, dim3((count + 255) / 256, rows), dim3(256), 0, s, in, pitch, out, q0, count);
}
size_t chan_lds_bytes(int M, int J)
{
    return (size_t)((CH_TI + J) * M + CH_TI * (M + 1) + M) * sizeof(float2) + (size_t)J * M * sizeof(float);
}
void launch_pfb_chan(const ChanParams& p, int batch, hipStream_t s)
{
    if (!p.m_count) return;
    for (const void* k : {reinterpret_cast<const void*>(k_pfb_chan<0>), reinterpret_cast<const void*>(k_pfb_chan<4>), reinterpret_cast<const void*>(k_pfb_chan<8>),
                          reinterpret_cast<const void*>(k_pfb_chan<12>), reinterpret_cast<const void*>(k_pfb_chan<16>)})
        if (dyn_lds_limit(k, 160 * 1024) != hipSuccess) return;
    // M = 64: the streaming kernel (rows 16-byte aligned, as qrl_chan_process demands of device buffers in practice); everything else -- other channel counts,
    // unaligned rows, ragged sample counts, and QRL_CHAN_OPT_LEGACY_PFB = 1 for the parity test of that path -- the general-M kernel.  (Round 3's tiled
    // 64-channel kernel k_pfb_chan64 was deleted in round 6: superseded, its A/B is on record in docs/KERNELS.md 5.)
    const bool aligned = (reinterpret_cast<uintptr_t>(p.in) & 15u) == 0 && (p.in_stride & 1u) == 0 && (p.n0 & 63u) == 0 && (p.n & 63u) == 0;
    if (p.M == 64 && p.J == 35 && p.legacy == 0 && aligned) {
        const auto kern = k_pfb_stream64<35>;
        if (dyn_lds_limit(reinterpret_cast<const void*>(kern), 160 * 1024) != hipSuccess) return;
        // segments: as many workgroups as the chip holds at once (occupancy x CUs), at least 4 tiles each
        static std::mutex mu; static int slots_cache[16];
        int dev = 0, slots;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return;
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!slots_cache[dev]) {
                int nb = 0; hipDeviceProp_t pr;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 256, stream64_lds_bytes()) != hipSuccess || nb < 1) nb = 1;
                if (const char* e = std::getenv("QRL_PFB_WG_PER_CU")) { const int v = std::atoi(e); if (v >= 1 && v < nb) nb = v; }   // experiment: fewer persistent workgroups per CU (room for the per-channel kernel beside them)
                slots_cache[dev] = nb * (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256);
            }
            slots = slots_cache[dev];
        }
        uint32_t nseg = (uint32_t)std::max(1, slots / batch);
        const uint32_t max_seg = (p.m_count + 4 * S64_T - 1) / (4 * S64_T);
        if (nseg > max_seg) nseg = max_seg;
        const uint32_t seg_len = ((p.m_count + nseg - 1) / nseg + S64_T - 1) / S64_T * S64_T;
        nseg = (p.m_count + seg_len - 1) / seg_len;
        hipLaunchKernelGGL(kern, dim3(nseg, batch), dim3(256), stream64_lds_bytes(), s, p, seg_len);
        return;
    }
    const dim3 grid((p.m_count + CH_TI - 1) / CH_TI, batch);
    const size_t lds = chan_lds_bytes(p.M, p.J);
    switch (p.M % 16 == 0 ? p.M / 4 : 0) {
    case 4:  hipLaunchKernelGGL(k_pfb_chan<4>, grid, dim3(256), lds, s, p); break;
    case 8:  hipLaunchKernelGGL(k_pfb_chan<8>, grid, dim3(256), lds, s, p); break;
    case 12: hipLaunchKernelGGL(k_pfb_chan<12>, grid, dim3(256), lds, s, p); break;
    case 16: hipLaunchKernelGGL(k_pfb_chan<16>, grid, dim3(256), lds, s, p); break;
    default: hipLaunchKernelGGL(k_pfb_chan<0>, grid, dim3(256), lds, s, p); break;
    }
}

__global__ __launch_bounds__(256) void k_f2s(const F2sParams P)
{
    const int b = blockIdx.y;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= P.count) return;
    const float x = P.in.p[(size_t)b * (P.in.mask + 1u) + ((uint32_t)(P.q0 + t) & P.in.mask)];
    float r = rintf((x * P.level) * P.scale);
    if (r > 32767.0f) r = 32767.0f;
    if (r < -32768.0f) r = -32768.0f;
    if (t < P.cap) P.out[(size_t)b * P.cap + t] = (int16_t)r;
    if (t == 0 && P.counts) P.counts[b] = P.count < P.cap ? P.count : (uint32_t)P.cap;
}
void launch_f2s(const F2sParams& p, int batch, hipStream_t s)
{
    if (!p.count) return;
    hipLaunchKernelGGL(k_f2s, dim3((p.count + 255) / 256, batch), dim3(256), 0, s, p);
}

// rssi_tag_block::work (reference src/gr/rssi_tag_block.cpp:43-68): every 300 samples one RSSI tag,
// 10 log10f(sqrtf(sum |x|^4 / 300) + 1e-20) + calibration, the sum being a serial float accumulation.  The 300-sample
// blocks sit on an absolute grid, so one thread per (stream, block) reproduces the serial sum exactly.
__global__ __launch_bounds__(64) void k_rssi_tag(const RssiParams P)
{
    const int b = blockIdx.y;
    const uint32_t t = blockIdx.x * 64u + threadIdx.x;
    if (t >= P.count) return;
    const uint64_t j = P.j0 + t;                       // absolute tag index
    const float2* ring = P.in.p + (size_t)b * (P.in.mask + 1u);
    float sum = 0.0f;
    for (int k = 0; k < 300; ++k) {
        const float2 x = ring[(uint32_t)(j * 300u + k) & P.in.mask];
        const float pwr = x.x * x.x + x.y * x.y;
        sum += pwr * pwr;
    }
    const float level = sqrtf(sum / 300.0f);
    const float db = 10.0f * log10f(level + 1.0e-20f) + P.calibration;
    if (t < P.cap) P.out[(size_t)b * P.cap + t] = db;
    if (t == 0 && P.counts) P.counts[b] = P.count < P.cap ? P.count : (uint32_t)P.cap;
}
void launch_rssi_tag(const RssiParams& p, int batch, hipStream_t s)
{
    if (!p.count) return;
    hipLaunchKernelGGL(k_rssi_tag, dim3((p.count + 63) / 64, batch), dim3(64), 0, s, p);
}

// ---- multi-carrier MMDVM transmitter (reference src/gr/gr_mod_mmdvm_multi2.cpp:30-128) ------------------------------------
// k_s2f_in   : short_to_float(1, 32767) + multiply_const_ff(level) into the float ring that feeds the FM modulator (k_tx_fm)
// k_scale_c  : multiply_const_cc(0.8) in place on the filtered ring
// k_pfb_synth: pfb_synthesizer_ccf(M = 10, taps, twox = false): per block the M port samples (ports = channel rings through
//              the {0,1,2,3,9,8,7} map, idle ports zero) go through the unnormalised inverse DFT (four real fmaf chains, p
//              ascending: the channelizer's contract), branch i is then filtered over its own history with h[i + M j] and the
//              M outputs leave in order, scaled by 1 / num_channels and the baseband gain.  A workgroup produces SY_TB blocks:
//              the IDFTs of SY_TB + J - 1 blocks are staged in LDS (the halo is recomputed from the channel rings).
__global__ __launch_bounds__(256) void k_s2f_in(const S2fInParams P)
{
    const int s = blockIdx.y;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= P.count) return;
    const float v = ((float)P.in[(size_t)s * P.in_stride + t] / P.scale) * P.level;
    P.out.p[(size_t)s * (P.out.mask + 1u) + ((uint32_t)(P.q0 + t) & P.out.mask)] = v;
}
void launch_s2f_in(const S2fInParams& p, int streams, hipStream_t s)
{
    if (!p.count) return;
    hipLaunchKernelGGL(k_s2f_in, dim3((p.count + 255) / 256, streams), dim3(256), 0, s, p);
}
__global__ __launch_bounds__(256) void k_scale_c(RingC r, uint64_t q0, uint32_t count, float k)
{
    const int s = blockIdx.y;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= count) return;
    float2* p = r.p + (size_t)s * (r.mask + 1u) + ((uint32_t)(q0 + t) & r.mask);
    float2 v = *p;
    v.x *= k; v.y *= k;
    *p = v;
}
void launch_scale_c(RingC r, uint64_t q0, uint32_t count, float k, int streams, hipStream_t s)
{
    if (!count) return;
    hipLaunchKernelGGL(k_scale_c, dim3((count + 255) / 256, streams), dim3(256), 0, s, r, q0, count, k);
}

// gr_zero_idle_bursts::work (reference src/gr/gr_zero_idle_bursts.cpp:45-84, delay 0): the items of a zero_samples run that fall
// into this call's range [lo, hi) of the ring are replaced by 0 + 0j.  One workgroup per run.
__global__ __launch_bounds__(256) void k_zero_runs(RingC r, const ZeroRun* runs, uint64_t lo, uint64_t hi)
{
    const ZeroRun z = runs[blockIdx.x];
    const uint64_t a = z.start > lo ? z.start : lo, b = z.start + z.count < hi ? z.start + z.count : hi;
    float2* row = r.p + (size_t)z.row * (r.mask + 1u);
    for (uint64_t i = a + threadIdx.x; i < b; i += 256) row[(uint32_t)i & r.mask] = make_float2(0.f, 0.f);
}
void launch_zero_runs(RingC r, const ZeroRun* runs, uint32_t nruns, uint64_t lo, uint64_t hi, hipStream_t s)
{
    if (!nruns || hi <= lo) return;
    hipLaunchKernelGGL(k_zero_runs, dim3(nruns), dim3(256), 0, s, r, runs, lo, hi);
}

constexpr int SY_TB = 96;   // output blocks per workgroup
__global__ __launch_bounds__(256) void k_pfb_synth(const SynthParams P)
{
    extern __shared__ __align__(16) unsigned char sy_smem[];
    const int M = P.M, J = P.J;
    float2* V = reinterpret_cast<float2*>(sy_smem);                 // [(SY_TB + J - 1)][M + 1]
    float* taps = reinterpret_cast<float*>(V + (SY_TB + J - 1) * (M + 1));   // J * M, zero padded
    float2* W = reinterpret_cast<float2*>(taps + J * M);            // M twiddles e^{+j 2 pi q / M}
    const int b = blockIdx.y, tid = threadIdx.x;
    const uint64_t blk0 = P.blk0 + (uint64_t)blockIdx.x * SY_TB;    // first output block of this workgroup (absolute)
    const int nb = (int)min((uint64_t)SY_TB, P.blk0 + P.nblk - blk0);
    for (int k = tid; k < J * M; k += 256) taps[k] = P.taps[k];
    for (int k = tid; k < M; k += 256) W[k] = P.twiddle[k];
    __syncthreads();
    // inverse DFT of blocks [blk0 - (J - 1), blk0 + nb): thread per (block, branch)
    const int nv = nb + J - 1;
    for (int w = tid; w < nv * M; w += 256) {
        const int r = w / M, i = w - r * M;
        const int64_t blk = (int64_t)blk0 - (J - 1) + r;
        float sa = 0.f, sb = 0.f, sc = 0.f, sd = 0.f;
        if (blk >= 0) {
            int q = 0;                                               // (i * p) mod M
            for (int p = 0; p < M; ++p) {
                const int ch = P.port_chan[p];                       // channel ring feeding port p, -1 = idle (null_source)
                float2 x = make_float2(0.f, 0.f);
                if (ch >= 0) x = P.in.p[((size_t)b * P.nch + ch) * (P.in.mask + 1u) + ((uint32_t)blk & P.in.mask)];
                const float2 wv = W[q];
                sa = fmaf(wv.x, x.x, sa); sb = fmaf(wv.y, x.y, sb); sc = fmaf(wv.y, x.x, sc); sd = fmaf(wv.x, x.y, sd);
                q += i; if (q >= M) q -= M;
            }
        }
        V[r * (M + 1) + i] = make_float2(sa - sb, sc + sd);
    }
    __syncthreads();
    // branch filters: out[(blk - P.blk0) M + i] = lvl * sum_j h[i + M j] V_i[blk - j]
    for (int w = tid; w < nb * M; w += 256) {
        const int r = w / M, i = w - r * M;
        const float2* vp = V + (r + J - 1) * (M + 1) + i;
        float ar = 0.f, ai = 0.f;
        for (int j = 0; j < J; ++j) {
            const float h = taps[i + M * j];
            const float2 v = vp[-j * (M + 1)];
            ar = fmaf(h, v.x, ar); ai = fmaf(h, v.y, ai);
        }
        ar *= P.level; ai *= P.level;
        ar *= P.bb_gain; ai *= P.bb_gain;
        const size_t o = (size_t)(blk0 - P.blk0 + r) * M + i;
        if (o < P.out_cap) P.out[(size_t)b * P.out_stride + o] = make_float2(ar, ai);
    }
}
size_t synth_lds_bytes(int M, int J) { return (size_t)((SY_TB + J - 1) * (M + 1) + M) * sizeof(float2) + (size_t)J * M * sizeof(float); }
void launch_pfb_synth(const SynthParams& p, int batch, hipStream_t s)
{
    if (!p.nblk) return;
    if (dyn_lds_limit(reinterpret_cast<const void*>(k_pfb_synth), 160 * 1024) != hipSuccess) return;
    hipLaunchKernelGGL(k_pfb_synth, dim3((p.nblk + SY_TB - 1) / SY_TB, batch), dim3(256), synth_lds_bytes(p.M, p.J), s, p);
}

}  // namespace qrl
