// kernels_chan_tail.hip — the per-channel chain of the multi-carrier MMDVM receiver as ONE kernel (gfx950 / CDNA4).
//
// Reference (src/gr/gr_demod_mmdvm_multi2.cpp:60-63,75-92,106-131), per 25 ksps channel behind the channelizer:
//   rational_resampler_ccf(24, 25, low_pass_2(1, 600k, 5k, 2k, 60, BH))  ->  fft_filter_ccf(low_pass_2(1, 24k, 5k, 2k, 60, BH))
//   -> rssi_tag_block -> quadrature_demod_cf(24000 / (2 pi 12500)) -> multiply_const_ff(level) -> float_to_short(1, 32767)
// and, for BASELINE config 4, the symbol demodulator's feed-forward part behind the same channel filter
// (src/gr/gr_demod_dmr.cpp:62-76): quadrature_demod_cf(24000 / (pi/2 4800)) -> fft_filter_fff(RRC(1, 24k, 4.8k, 0.2, 125)).
//
// Round 2 ran this as k_resamp, k_fir_ccf_tiled, k_rssi_tag, k_quad_demod, k_fir_fff_tiled: 4.4 ms per call of 134 M wideband samples,
// every stage through HBM-sized rings, one LDS read of a tap and one of a sample per FMA pair.  Here a workgroup owns one channel
// and a tile of 1200 outputs at 24 ksps (tiles on an ABSOLUTE grid, = 4 RSSI blocks of 300):
//   stage the 25 ksps input span once -> A resampler -> B channel filter -> D discriminators (+ int16) -> E RRC, all in LDS, the
//   serial 300-sample RSSI sums (C) on otherwise idle lanes beside E.  Halos (171 outputs in front of the tile) are recomputed
//   from the channel ring instead of being read back from intermediate rings: every intermediate value is a pure function of the
//   ring, so the recomputed ones are the bits the previous tile / call produced.
// Register blocking: a thread computes 8 CONSECUTIVE outputs of a filter over a sliding register window -- one LDS read of a sample
// per 8 FMA pairs -- and the taps are wave-uniform LDS broadcast reads out of step-major tables (step s = 7 - d, d = r - k: the 8 taps
// h[r - d], r = 0..7, of a step are 32 contiguous bytes, zero where r - d falls outside the filter: a zero tap leaves the chain
// untouched, fmaf(0, x, acc) = acc).  The step loops are rolled (unroll 4): fully unrolled, the compiler hoists every tap read
// to the top of the stage (256 VGPRs or 500 scratch reloads); as scalar loads the 280 resampler taps of a wave overflow the SGPR file.  The LDS images are de-interleaved by 8 (item i at
// (i & 7) W + (i >> 3), W = 4 mod 32) so that "thread g reads item 8 g + d" is lane-contiguous and "thread t reads item t" still
// spreads over all banks.  The resampler (24 phases) maps lane = group of 24 outputs, wave = which 8 of them: the phase of an output
// is then wave-uniform and c(q) = floor(25 q / 24) has no carry inside the 8 (x index 25 u + 8 w + r - j, lane stride 25 samples:
// conflict free for ds_read_b64).
// Round 5: stages B (33-tap channel filter) and E (125-tap RRC) run on the f32 MATRIX pipe as Toeplitz products (v_mfma_f32_16x16x4_f32: rows = 16
// consecutive outputs, columns = blocks of 16 outputs, K = four sample offsets in descending order = tap index ascending): see the two stages.  The
// register-blocked packed-fma form described above is what stage A (the resampler) still uses.
// Every chain is the oracle's: one fmaf chain per output, tap index ascending, first term fmaf(h, x, +0) (orc_resamp_ccf,
// orc_fir_ccf, orc_fir_fff); discriminator, quantiser and RSSI sums as in k_quad_demod / k_rssi_tag.
#include <cstring>
#include <vector>
#include "devmath.hpp"
#include "engine.hpp"

namespace qrl {

constexpr int CT_T = 1200;          // outputs per tile (absolute grid)
#ifndef QRL_CT_TPW
#define QRL_CT_TPW 1
#endif
constexpr int CT_TPW = QRL_CT_TPW;           // consecutive tiles of a row one workgroup walks (3 was measured in round 4: staging the 10.5 KB of tables once per three tiles changes nothing, 1859 against 1867 us)
constexpr int CT_JP = 35;           // taps per phase of the 24/25 resampler (819 taps)
constexpr int CT_NF = 33;           // channel filter
constexpr int CT_NR = 125;          // RRC
constexpr int CT_HA = 171;          // outputs in front of the tile the resampler produces (>= 132 + 7 + 32)
constexpr int CT_HB = 132;          // ... the channel filter / discriminator produce (>= 124 + 7 + 1)
constexpr int CT_W = 180;           // row pitch of the de-interleaved images: >= 24 * 59 / 8, and (l & 7) W + (l >> 3), l < 32, hits 32 different bank pairs (W = 4 x odd)
constexpr int CT_NX = 25 * 59 + 34 + 2;   // input samples staged per tile
constexpr int CT_SA = CT_JP + 7, CT_SB = CT_NF + 7;   // window steps of the 8-output sliding filters (d = 7 .. -(nt - 1))
// stage E on the matrix pipe (round 5): the RRC as a Toeplitz product, v_mfma_f32_16x16x4_f32 = rows: 16 consecutive outputs, columns: 16 blocks
// of 16 outputs, K: 4 sample offsets d (see the stage).  CT_EM matrix instructions cover d = 15 .. -(CT_NR - 1); the discriminator image is stored
// item i at (i & 15) CT_DP + (i >> 4) so that "lane (kk, n) reads item 16 n + c - kk" spreads over the banks (CT_DP = 17 mod 32).
constexpr int CT_EM = (CT_NR + 15 + 3) / 4;            // 35
// stage B on the matrix pipe as well (round 5): rows = 16 consecutive filter outputs, columns = 8 blocks of 16 outputs x (re, im), K = 4 sample offsets;
// CT_BM matrix instructions cover d = 15 .. -(CT_NF - 1); the padded tap table of the stage has 15 zeros in front (tB[j] = ft[j - 15])
constexpr int CT_BM = (CT_NF + 15 + 3) / 4;            // 12
constexpr int CT_HBT = 64;                             // >= 15 + 3 + 4 CT_BM
constexpr int CT_DP = 113;                             // >= (139 + 16 * 79 + 15) / 16 + 1 = 89 columns
constexpr int CT_HE = 160;                             // padded tap table: hpad[j] = rrc[j - 15], j < 15 + 3 + 4 CT_EM
typedef float f32x4_ct __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int ct_dpos(int i) { return (i & 15) * CT_DP + (i >> 4); }

#ifdef QRL_CT_PROF
// developer build (tools/chan_tail_variants.sh name -DQRL_CT_PROF): shader-clock ticks per phase, summed over every wave
__device__ unsigned long long g_ct_prof[4096][16];   // spread over 4096 slots: one shared set of counters serialises 2 M waves on a few cache lines
#define CT_STAMP(k) do { const unsigned long long tn_ = __builtin_readcyclecounter(); pc[k] += tn_ - tprev; tprev = tn_; } while (0)
#else
#define CT_STAMP(k) do { } while (0)
#endif
__device__ __forceinline__ int ct_pos(int i) { return (i & 7) * CT_W + (i >> 3); }
// Packed fmas (v_pk_fma_f32: two IEEE fmas per instruction, the rounding of fmaf): complex accumulator += real tap x complex sample with
// the tap broadcast from one half of a register PAIR through op_sel -- the eight taps of a step arrive as two float4 = four pairs, so no
// register is spent on duplicating them.  Halves the issue count of the three filter stages (they are VALU-issue bound).
typedef float v2f_ct __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void ct_fma_lo(v2f_ct& acc, v2f_ct hp, v2f_ct x) { asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(hp), "v"(x)); }
__device__ __forceinline__ void ct_fma_hi(v2f_ct& acc, v2f_ct hp, v2f_ct x) { asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(hp), "v"(x)); }
__device__ __forceinline__ void ct_step8(v2f_ct (&acc)[8], float4 h0, float4 h1, float2 xs)
{
    const v2f_ct x = {xs.x, xs.y}, p0 = {h0.x, h0.y}, p1 = {h0.z, h0.w}, p2 = {h1.x, h1.y}, p3 = {h1.z, h1.w};
    ct_fma_lo(acc[0], p0, x); ct_fma_hi(acc[1], p0, x); ct_fma_lo(acc[2], p1, x); ct_fma_hi(acc[3], p1, x);
    ct_fma_lo(acc[4], p2, x); ct_fma_hi(acc[5], p2, x); ct_fma_lo(acc[6], p3, x); ct_fma_hi(acc[7], p3, x);
}
__device__ __forceinline__ int64_t ct_floordiv(int64_t a, int64_t b) { const int64_t q = a / b; return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q; }

#ifndef QRL_CT_WPE
#define QRL_CT_WPE 4     // waves per SIMD the register allocation aims at.  5 (96 VGPRs, a few loop-invariant values in scratch: a SIMD then holds four waves of
                         // this kernel AND one of the previous call's symbol synchroniser, and the LDS -- 32.5 KB -- allows it) was measured in round 5:
                         // slower, 2.98 - 3.03 against 2.86 - 2.92 ms per C4 step (profiles/r05_c4_tail_experiments.log)
#endif
__global__ __launch_bounds__(256, QRL_CT_WPE) void k_chan_tail(const ChanTailParams P)
{
    __shared__ __align__(16) float2 xf[CT_NX > 8 * CT_W ? CT_NX : 8 * CT_W];   // staged input x (stage A), then the channel filter output f (B .. E)
    __shared__ __align__(16) float2 av[8 * CT_W];        // resampler output a
    __shared__ __align__(16) float dv[16 * CT_DP];       // symbol discriminator output d2 (ct_dpos); stages A / B: their tap tables
    __shared__ float T[257];
    __shared__ __align__(16) float tE[CT_HE];            // RRC taps, 15 zeros in front (stage E)
    // [step][r] tap tables of stages A and B live where stage D will put the discriminator image (round 5: 32.5 KB instead of 39.9 KB per
    // workgroup, so that FIVE workgroups' worth of LDS is there: four of this kernel + the symbol synchroniser of the previous call)
    float* const tA = dv; float* const tB = dv + 3 * CT_SA * 8;
    static_assert(3 * CT_SA * 8 + CT_SB * 8 <= 16 * CT_DP, "tap tables do not fit the discriminator image");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = blockIdx.y;
#ifdef QRL_CT_PROF
    unsigned long long pc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_readcyclecounter();
#endif
    const int64_t tile_first = (int64_t)(P.q0 / CT_T) + (int64_t)blockIdx.x * CT_TPW, tile_last = (int64_t)((P.q0 + P.count - 1) / CT_T);
    const int ntile = (int)(tile_last - tile_first + 1 < CT_TPW ? tile_last - tile_first + 1 : CT_TPW);
    for (int k = tid; k < 257; k += 256) T[k] = P.atan_tab[k];
    if (P.out_sym.p && tid < CT_HE) tE[tid] = P.tab_e[tid];
    for (int tt = 0; tt < ntile; ++tt) {
    const int64_t tile = tile_first + tt;
    const int64_t Q0 = tile * CT_T;
    if (tt) __syncthreads();                                                      // the previous tile's readers of xf / av / dv are through
    // step-major tap tables, laid out by the host (chan_tail_tables): straight 16-byte copies (per tile: stage D overwrites them)
    for (int k = tid; k < 3 * CT_SA * 2; k += 256) reinterpret_cast<float4*>(tA)[k] = reinterpret_cast<const float4*>(P.tab_a)[k];
    if (tid < CT_HBT) tB[tid] = P.tab_b[tid];
    CT_STAMP(0);
    const int64_t ua = ct_floordiv(Q0 - CT_HA, 24), qa = ua * 24;                 // first resampler output of the tile (multiple of 24)
    const int NU = (int)((Q0 + CT_T - qa + 23) / 24);                             // groups of 24 outputs: <= 59
    const int64_t qb = qa + ((Q0 - CT_HB - qa) / 8) * 8;                          // first filter output (multiple of 8 behind qa)
    const int ib0 = (int)(qb - qa);                                               // >= 32
    const int NB = (int)(Q0 + CT_T - qb);                                         // filter outputs: 1332 .. 1339
    const int e0 = (int)(Q0 - qb);                                                // tile start relative to qb: 132 .. 139
    // ---- stage the input: x[xbase + i], xbase = 25 ua - 34 (zero in front of the stream)
    const int64_t xbase = 25 * ua - (CT_JP - 1);
    const int nx = 25 * NU + (CT_JP - 1);
    {   // all loads of the thread are issued before the first one is used (ring reads are in bounds for any index)
        const float2* ring = P.in.p + (size_t)row * (P.in.mask + 1u);
        constexpr int NLD = (CT_NX + 255) / 256;
        float2 v[NLD];
        if (P.lin) {
            // form 3: items of this call straight from the caller's rows, older ones from the history rows (reads behind the call's last item are
            // clamped: they only feed outputs that do not exist yet)
            const float2* lrow = P.lin + (size_t)row * P.lin_pitch;
            const float2* hrow = P.hist + (size_t)row * P.hist_len;
            const int64_t rel0 = xbase - (int64_t)P.lin_base;
            if (rel0 >= 0 && rel0 + CT_NX + 256 <= (int64_t)P.lin_n) {                 // the whole span lies in this call's rows: every tile but a call's first and last
#pragma unroll
                for (int k = 0; k < NLD; ++k) v[k] = lrow[(size_t)rel0 + tid + 256 * k];
            } else
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const int64_t a = xbase + tid + 256 * k;
                const int64_t rel = a - (int64_t)P.lin_base;
                const uint32_t rl = rel < 0 ? 0u : ((uint64_t)rel < P.lin_n ? (uint32_t)rel : P.lin_n - 1u);
                const int64_t hi = rel + (int64_t)P.hist_len;
                const uint32_t hl = hi < 0 ? 0u : (hi < (int64_t)P.hist_len ? (uint32_t)hi : P.hist_len - 1u);
                const float2* src = rel >= 0 ? lrow + rl : hrow + hl;              // ONE load per item (a select of two loaded values doubled the kernel's global loads)
                v[k] = *src;
                if (a < 0 || hi < 0) v[k] = make_float2(0.f, 0.f);
            }
        } else {
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int64_t a = xbase + tid + 256 * k;
            v[k] = ring[(uint32_t)a & P.in.mask];
            if (a < 0) v[k] = make_float2(0.f, 0.f);
        }
        }
#pragma unroll
        CT_STAMP(1);
#pragma unroll
        for (int k = 0; k < NLD; ++k) { const int i = tid + 256 * k; if (i < nx) xf[i] = v[k]; }
    }
    CT_STAMP(2);
    __syncthreads();
    CT_STAMP(3);
    // ---- A: a[24 u + 8 w + r] = sum_j taps[(8 w + r) 35 + j] x[25 u + 8 w + r - j],  lane = u - ua, waves 0..2
    if (wv < 3 && lane < NU) {
        const float4* tp = reinterpret_cast<const float4*>(tA + wv * (CT_SA * 8));
        const float2* xb = xf + 25 * lane + 8 * wv + (CT_JP - 1) + 7;            // x[25 u + 8 w + d] = xb[d - 7] = xb[-s]
        v2f_ct acc[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) acc[r] = v2f_ct{0.f, 0.f};
#pragma unroll 4
        for (int st = 0; st < CT_SA; ++st) ct_step8(acc, tp[2 * st], tp[2 * st + 1], xb[-st]);
#pragma unroll
        for (int r = 0; r < 8; ++r) av[r * CT_W + 3 * lane + wv] = make_float2(acc[r].x, acc[r].y);   // item i = 24 lane + 8 w + r
    }
    // the matrix forms of stages B and E multiply ZERO taps with up to 15 items BEHIND an output, and a zero tap leaves the chain untouched only if
    // the sample is finite (0 x NaN = NaN): the items behind the last resampler output must not be whatever the LDS held before (found by
    // tests/test_gpu_sharding.py: the last outputs of a tile differed when the memory behind the image happened to hold a NaN pattern)
    if (tid < 32) av[ct_pos(24 * NU + tid)] = make_float2(0.f, 0.f);
    CT_STAMP(4);
    __syncthreads();
    CT_STAMP(5);
    // ---- B: f[q] = sum_k ft[k] a[q - k], thread g: q = qb + 8 g + r (items of a: ib0 + 8 g + r - k)
    // The kernel is LDS bound (two 16-byte tap reads + one sample read per step against 8 packed fmas: halving the fma count
    // changed nothing), and this stage is the largest: here the 33 taps live in REGISTERS (17 pairs, read once per tile out of rows
    // 7, 15, 23, 31, 39 of the step-major table: row 8 i + 7 holds h[8 i .. 8 i + 7]), the 40 steps are unrolled with compile-time
    // tap indices -- one LDS read (the sample) per step, and the taps that are zero padding are not multiplied at all.
    {
        // Round 5: the same Toeplitz form as stage E.  Output i' = 16 n + i (relative to qb), component c:  F[i][(n, c)] = sum_d A[i][d] B[d][(n, c)],
        // A[i][d] = ft[i - d] (0 outside the filter), B[d][(n, c)] = component c of a[ib0 + 16 n + d], d = 15 .. -32 in DESCENDING order, four per
        // v_mfma_f32_16x16x4_f32 (slot kk of instruction m: d = 15 - 4 m - kk, tap index i - 15 + 4 m + kk): per component the oracle's chain
        // fmaf(ft[k], a[q - k], acc), k ascending from +0 (orc_fir_ccf), bit for bit.  16 columns = 8 blocks x (re, im): a chain = 128 complex outputs in
        // CT_BM = 12 instructions; 11 chains cover the NB <= 1339 outputs of a tile (3 + 3 + 3 + 2 over the four waves).  The resampler image `av`
        // keeps its layout (item at (item & 7) W + (item >> 3)): the item walks down by 4 per instruction = the other row of its pair, one column
        // every second step -- two base addresses and immediate offsets; results go to the f image in the layout stage D reads.  (The 40 tap
        // registers of the packed-fma form were the kernel's register peak: 128 -> 103 VGPRs.)
        const int col = lane & 15, kk = lane >> 4, cmp = col & 1, nl = col >> 1;
        const float* avf = reinterpret_cast<const float*>(av);
        float* xff = reinterpret_cast<float*>(xf);
        const float* hp = tB + col + kk;                                          // A operand: row i = lane & 15, slot kk: tB[i + kk + 4 m]
#pragma unroll 1
        for (int cg = wv; cg < (NB + 127) / 128; cg += 4) {
            const int n = 8 * cg + nl;
            const int it0 = ib0 + 16 * n + 15 - kk;
            int ob[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) ob[j] = 2 * (ct_pos(it0 - 4 * j) - 6) + cmp;
            f32x4_ct acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < CT_BM; ++m) {
                const float a_ = hp[4 * m];
                const float b_ = avf[ob[m & 1] + 2 * (6 - (m >> 1))];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_, b_, acc, 0, 0, 0);
            }
            // lane holds outputs i' = 16 n + 4 kk + r, r = 0..3, of component cmp
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i2 = 16 * n + 4 * kk + r;
                xff[2 * ct_pos(i2) + cmp] = acc[r];
            }
        }
    }
    CT_STAMP(6);
    __syncthreads();
    CT_STAMP(7);
    // ---- D: discriminators on f, items i' = 1 .. NB - 1 (q = qb + i'); int16 port for the outputs of this call
    const uint64_t q_end = P.q0 + P.count;
    float* pv = reinterpret_cast<float*>(av);
    {   // a thread's (at most six) items as ONE straight-line block: all LDS reads first, then six independent discriminator chains the
        // scheduler can interleave (as a loop over i the stage waited for the LDS and the divide of every item in turn: 6.5 k cycles)
        constexpr int ND = (CT_T + CT_HB + 8 + 255) / 256;                         // NB <= 1339
        float2 aa[ND], pp[ND];
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            const int i = 1 + tid + 256 * j, ic = i < NB ? i : 1;
            aa[j] = xf[ct_pos(ic)]; pp[j] = xf[ct_pos(ic - 1)];
        }
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            const int i = 1 + tid + 256 * j;
            const float2 a = aa[j], p = pp[j];
            const float re = a.x * p.x + a.y * p.y;
            const float im = a.y * p.x - a.x * p.y;
            const float ang = fast_atan2f_lut(im, re, T);
            if (i < NB) dv[ct_dpos(i)] = P.gain2 * ang;
            if (j == 0 && tid < 16) dv[ct_dpos(NB + tid)] = 0.0f;                 // (stage E: finite items behind the last one, as for stage B)
            if (i < NB && i >= e0) {   // |f|^4 of the tile's own items for the serial RSSI sums below, in item order (the resampler output `av` is dead since stage B: its memory holds them)
                const float pwr = a.x * a.x + a.y * a.y;
                pv[i - e0] = pwr * pwr;
            }
            const int64_t q = qb + i;
            if (i < NB && P.s16 && q >= (int64_t)P.q0 && (uint64_t)q < q_end && i >= e0) {
                float r = rintf(((P.gain * ang) * P.level) * P.scale);
                if (r > 32767.0f) r = 32767.0f;
                if (r < -32768.0f) r = -32768.0f;
                const uint64_t t = (uint64_t)q - P.q0;
                if (t < P.s16_cap) P.s16[(size_t)row * P.s16_cap + t] = (int16_t)r;
            }
        }
    }
    if (blockIdx.x == 0 && tt == 0 && tid == 0) {
        if (P.s16 && P.s16_counts) P.s16_counts[row] = P.count < P.s16_cap ? P.count : (uint32_t)P.s16_cap;
        if (P.rssi && P.rssi_counts) P.rssi_counts[row] = P.ntags < P.rssi_cap ? P.ntags : (uint32_t)P.rssi_cap;
    }
    CT_STAMP(8);
    __syncthreads();
    CT_STAMP(9);
    // ---- E: r[q] = sum_k rrc[k] d2[q - k] on the MATRIX pipe.  Output o = 16 n + i of the tile (item e0 + o of the d2 image): a Toeplitz product
    //   Y[i][n] = sum_d A[i][d] B[d][n],  A[i][d] = rrc[i - d] (0 outside the filter),  B[d][n] = d2[e0 + 16 n + d],  d = 15 .. -124,
    // four d per v_mfma_f32_16x16x4_f32 in DESCENDING order (slot kk of instruction m: d = 15 - 4 m - kk, tap index i - 15 + 4 m + kk): the
    // instruction adds its four products to the accumulator one after the other with one rounding each (the pm contract of the front ends,
    // DESIGN 2), so an output is the oracle's chain -- fmaf(rrc[k], d2[q - k], acc), k ascending from +0 -- bit for bit: a zero tap leaves the
    // chain untouched.  One chain = 16 x 16 outputs x 125 taps in CT_EM = 35 matrix instructions (89 % of their MACs are taps); the 75 blocks of a
    // tile are 5 chains: waves 0..2 take one each and waves 0 / 1 a second one, wave 3 the serial RSSI sums.  Per instruction two 4-byte LDS
    // reads with immediate offsets (A: tE[i + kk + 4 m]; B: the item walks down by 4 = one row of four, a column every fourth step) and no VALU
    // work -- the scalar-fma form of round 4 issued 1056 fmas per thread for 8 outputs and was 48 % of the kernel's VALU instructions.
    auto e_chain = [&](int cg) {
        const int n = 16 * cg + (lane & 15), kk = lane >> 4;
        const int it0 = e0 + 16 * n + 15 - kk;                                    // item of instruction m = 0
        int ob[4];                                                                // LDS index of item it0 - 4 j, minus the 8 columns the walk can go back
#pragma unroll
        for (int j = 0; j < 4; ++j) ob[j] = ct_dpos(it0 - 4 * j) - 8;
        const float* hp = tE + (lane & 15) + kk;
        f32x4_ct acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < CT_EM; ++m) {
            const float a = hp[4 * m];
            const float b = dv[ob[m & 3] + 8 - (m >> 2)];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
        }
        if (n < CT_T / 16) {
            float* orow = P.out_sym.p + (size_t)row * (P.out_sym.mask + 1u);
            const int64_t q4 = Q0 + 16 * n + 4 * kk;                              // lane holds outputs q4 .. q4 + 3 (rows 4 kk + r of column n)
            if (q4 >= (int64_t)P.q0 && (uint64_t)(q4 + 3) < q_end)
                *reinterpret_cast<float4*>(orow + ((uint32_t)q4 & P.out_sym.mask)) = make_float4(acc[0], acc[1], acc[2], acc[3]);   // q4 = 0 mod 4: inside one ring row
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t q = q4 + r;
                    if (q >= (int64_t)P.q0 && (uint64_t)q < q_end) orow[(uint32_t)q & P.out_sym.mask] = acc[r];
                }
            }
        }
    };
    if (wv == 3) {
        // ---- C: rssi_tag_block, the four 300-item blocks of this tile: serial float sums, one lane per block
        if (lane < CT_T / 300 && P.rssi) {
            const int64_t j = tile * (CT_T / 300) + lane;                          // absolute tag index
            if (j >= (int64_t)P.tag0 && j < (int64_t)(P.tag0 + P.ntags)) {
                // (16-byte reads of the block's 300 values, eight in flight: one 4-byte LDS read per add made this wave the longest of the
                //  tile -- 28.7 k of 54 k cycles, profiles/r04_c4_chan_tail_phase_profile_before.log)
                const float4* p4 = reinterpret_cast<const float4*>(pv + 300 * lane);
                float sum = 0.0f;                                                  // the block's serial sum, item order (rssi_tag_block.cpp:52-58)
                for (int k0 = 0; k0 < 75; k0 += 5) {
                    float4 v[5];
#pragma unroll
                    for (int u = 0; u < 5; ++u) v[u] = p4[k0 + u];
#pragma unroll
                    for (int u = 0; u < 5; ++u) { sum += v[u].x; sum += v[u].y; sum += v[u].z; sum += v[u].w; }
                }
                const float level = sqrtf(sum / 300.0f);
                const float db = 10.0f * log10f(level + 1.0e-20f) + P.rssi_cal;
                const uint64_t t = (uint64_t)j - P.tag0;
                if (t < P.rssi_cap) P.rssi[(size_t)row * P.rssi_cap + t] = db;
            }
        }
    }
    if (P.out_sym.p) {
        // chains of this wave: its own (wave 3 only when it has no RSSI sums to do), then chain 3 (wave 0, when wave 3 is busy) / chain 4 (wave 1)
        const int first = (wv < 3 || !P.rssi) ? wv : 5, second = (wv == 0 && P.rssi) ? 3 : (wv == 1 ? 4 : 5);
#pragma unroll 1
        for (int c = 0; c < 2; ++c) {
            const int cg = c ? second : first;
            if (cg < 5) e_chain(cg);
        }
    }
    CT_STAMP(10);
    }   // tiles of this workgroup
#ifdef QRL_CT_PROF
    if (lane == 0) {
        unsigned long long* slot = g_ct_prof[(blockIdx.x * 61u + blockIdx.y * 7u + (unsigned)wv) & 4095u];
        for (int k = 0; k < 11; ++k) atomicAdd(&slot[k], pc[k]);
        atomicAdd(&slot[11 + (wv == 3)], pc[10]); atomicAdd(&slot[15], 1ull);
    }
#endif
}
#ifdef QRL_CT_PROF
extern "C" void qrl_ct_prof_read(unsigned long long* out16)
{
    (void)hipDeviceSynchronize();
    static unsigned long long host[4096][16];
    (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ct_prof), sizeof host);
    for (int k = 0; k < 16; ++k) { out16[k] = 0; for (int i = 0; i < 4096; ++i) out16[k] += host[i][k]; }
    std::memset(host, 0, sizeof host);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ct_prof), host, sizeof host);
}
#endif

void launch_chan_tail(const ChanTailParams& p, int streams, hipStream_t s)
{
    if (!p.count) return;
    const uint64_t t_first = p.q0 / CT_T, t_last = (p.q0 + p.count - 1) / CT_T;
    hipLaunchKernelGGL(k_chan_tail, dim3((uint32_t)((t_last - t_first + CT_TPW) / CT_TPW), streams), dim3(256), 0, s, p);
}
// step-major tap tables of k_chan_tail: which = 0: resampler [3 waves][42 steps][8] from the phase-major taps[24][35];
// 1: channel filter [40][8]; entry (step s, r) = h[r - (7 - s)], zero outside the filter.  2: RRC for stage E, hpad[160] (15 zeros, the taps, zeros).
std::vector<float> chan_tail_tables(int which, const float* taps)
{
    if (which == 0) {
        std::vector<float> t((size_t)3 * CT_SA * 8, 0.0f);
        for (int w = 0; w < 3; ++w)
            for (int st = 0; st < CT_SA; ++st)
                for (int r = 0; r < 8; ++r) {
                    const int j = r - (7 - st);
                    if (j >= 0 && j < CT_JP) t[((size_t)w * CT_SA + st) * 8 + r] = taps[(8 * w + r) * CT_JP + j];
                }
        return t;
    }
    if (which == 2) {   // stage E (matrix pipe): hpad[j] = rrc[j - 15]
        std::vector<float> t((size_t)CT_HE, 0.0f);
        for (int k = 0; k < CT_NR; ++k) t[15 + k] = taps[k];
        return t;
    }
    // which == 1: stage B (matrix pipe): tB[j] = ft[j - 15]
    std::vector<float> t((size_t)CT_HBT, 0.0f);
    for (int k = 0; k < CT_NF; ++k) t[15 + k] = taps[k];
    return t;
}
bool chan_tail_supported(int rs_I, int rs_D, int rs_Jp, int filt_nt, int rrc_nt)
{
    return rs_I == 24 && rs_D == 25 && rs_Jp == CT_JP && filt_nt == CT_NF && (rrc_nt == 0 || rrc_nt == CT_NR);
}
uint32_t chan_tail_lookback() { return (uint32_t)((CT_T + CT_HA + 24) * 25 / 24 + CT_JP + 64); }   // channel-ring items in front of a call the tiles may re-read

}  // namespace qrl
