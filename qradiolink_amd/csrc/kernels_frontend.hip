// kernels_frontend.hip — the HBM-facing kernels of the RX path (gfx950 / CDNA4).
//
//  k_decim  : rotator_cc + rational_resampler_ccf(1, D, taps)       [gr_demod_base.cpp:57,1330-1340;
//             gr_demod_2fsk.cpp:82-88; gr_demod_qpsk.cpp:92-96]
//  k_resamp : rational_resampler_ccf(I, D, taps), I > 1             [gr_demod_gmsk.cpp:80-83]
//  k_hist   : keeps the last H rotated input samples of each stream for the next call
//
// k_decim layout.  One workgroup = 4 waves = the 4 phase groups of the summation contract
// (oracle/orc_blocks.c orc_decim_fir_ccf): wave g owns phases p in [gD/4, (g+1)D/4) and forms ONE
// fmaf chain per output (p ascending, j ascending); the four partial sums meet in LDS and are
// combined as (r0+r1)+(r2+r3).  A tile of TILE = 64*R outputs needs input blocks
// bq in [mt-Jpad, mt+TILE): they are read from HBM once, coalesced (16 B per lane), rotated on the
// way in (phase = T_hi[k>>9] (x) T_lo[k&511]) and stored to LDS TRANSPOSED: row o = sample index
// mod D, column = input block.  In that layout tap (p, j) of output m sits at row (D-p)%D, column
// m-j-(p>0): a lane that owns R consecutive outputs slides a register window of R+JC-1 columns over
// JC taps (R*JC packed FMAs per R+JC-1 LDS reads), and the columns are de-interleaved by R
// (pos = (col%R)*Wq + col/R) so that the 64 lanes of a wave hit consecutive 8-byte LDS words.
// Taps are wave-uniform => scalar loads.  blockIdx -> tile mapping keeps neighbouring tiles (which
// share the Jpad-block halo) on the same XCD/L2.
#include "devmath.hpp"
#include "engine.hpp"

namespace qrl {

typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float2 rot_apply(float2 x, uint64_t k, const float2* t_hi, uint32_t kb0, const float2* t_lo)
{
    const float2 hi = t_hi[(uint32_t)(k >> 9) - kb0];
    const float2 lo = t_lo[(uint32_t)k & 511u];
    return cmul_fma(x, cmul_fma(hi, lo));
}
// krel = k - (kb0 << 9) fits 32 bits inside a tile
__device__ __forceinline__ float2 rot_apply_rel(float2 x, uint32_t krel, const float2* t_hi, const float2* t_lo)
{
    return cmul_fma(x, cmul_fma(t_hi[krel >> 9], t_lo[krel & 511u]));
}

// fetch one input sample of stream b at absolute index i (zero outside the stream so far)
__device__ __forceinline__ float2 decim_fetch(const DecimParams& P, int b, int64_t i, const float2* t_hi, uint32_t kb0,
                                              const float2* t_lo)
{
    if (i < 0) return make_float2(0.f, 0.f);
    const uint64_t ui = (uint64_t)i;
    if (P.in) {
        if (ui >= P.n0 + P.n) return make_float2(0.f, 0.f);
        if (ui >= P.n0) {
            float2 x = P.in[(size_t)b * P.in_stride + (size_t)(ui - P.n0)];
            if (P.rot_enable) x = rot_apply(x, ui - P.rot_nbase, t_hi, kb0, t_lo);
            return x;
        }
        const uint64_t d = P.n0 - ui;
        if (d > P.hist_len) return make_float2(0.f, 0.f);
        return P.hist[(size_t)b * P.hist_len + (P.hist_len - (uint32_t)d)];
    }
    if (ui >= P.n0 + P.n) return make_float2(0.f, 0.f);
    return P.in_ring.p[(size_t)b * (P.in_ring.mask + 1u) + ((uint32_t)ui & P.in_ring.mask)];
}

template <int R, int JC>
__global__ __launch_bounds__(256) void k_decim(const DecimParams P_)
{
    const DecimParams& P = P_;
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int TILE = 64 * R;
    const int D = P.D, Jpad = P.Jpad;
    const int W = TILE + Jpad;
    const int Wq = (W + R - 1) / R;
    const int PT = (R * Wq) | 1;                      // odd row pitch: <=2-way conflicts on the transposing stores
    float2* t_lo = reinterpret_cast<float2*>(smem);   // 512
    float2* t_hi = t_lo + 512;                        // 64
    float2* part = t_hi + 64;                         // 4 * TILE
    float2* tile = part + 4 * TILE;                   // D * PT

    const int b = blockIdx.y;
    const uint32_t per = gridDim.x >> 3;
    const uint32_t tix = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if (tix >= P.tiles) return;
    const int tid = threadIdx.x;
    const uint64_t mt = P.m0 + (uint64_t)tix * TILE;
    const int64_t bq0 = (int64_t)mt - Jpad;
    const int64_t i_start = bq0 * D;
    const int nsamp = W * D;

    // ---- rotator tables for this tile ----
    uint32_t kb0 = 0;
    if (P.rot_enable) {
        t_lo[tid] = P.rot_lo[tid];
        t_lo[tid + 256] = P.rot_lo[tid + 256];
        const int64_t first_new = i_start > (int64_t)P.n0 ? i_start : (int64_t)P.n0;
        kb0 = (uint32_t)(((uint64_t)first_new - P.rot_nbase) >> 9);
        if (tid < 64) t_hi[tid] = sincos_turn(P.rot_acc + ((uint64_t)(kb0 + tid) << 9) * P.rot_inc);
        __syncthreads();
    }

    // ---- stage the tile: coalesced 16-byte loads, rotate, transposed LDS store ----
    // Stored samples: i = i_first + s, s in [0, ns), i_first = i_start + 1 (sample (o=0, bq0) is never
    // used).  o = (s+1) % D, cb = (s+1) / D, column = cb - (o == 0): row 0 sits one column to the left so
    // that every phase reads column m-j-1 (see the FIR loop).
    {
        const int64_t i_first = i_start + 1;
        const int ns = nsamp - 1;
        const int a = (int)((i_first - (int64_t)P.n0) & 1);  // pairs start at s = -a so that (i - n0) is even
        // pairs k in [k_lo, k_hi): both samples inside the caller's buffer and inside the tile
        int k_lo = 0, k_hi = 0;
        if (P.in) {
            int64_t lo = ((int64_t)P.n0 - i_first + a + 1) >> 1;                 // i0 >= n0
            int64_t hi = ((int64_t)(P.n0 + P.n) - i_first + a - 1) >> 1;         // i0 + 1 < n0 + n
            if (lo < a) lo = a;                                                  // s0 >= 0
            if (hi > ((ns + a) >> 1)) hi = (ns + a) >> 1;                        // s0 + 1 < ns
            if (hi < lo) hi = lo;
            k_lo = (int)lo; k_hi = (int)hi;
        }
        const int s_lo = k_hi > k_lo ? 2 * k_lo - a : 0, s_hi = k_hi > k_lo ? 2 * k_hi - a : 0;
        // checked single samples: [0, s_lo) and [s_hi, ns)
        for (int seg = 0; seg < 2; ++seg) {
            const int sb = seg ? s_hi : 0, se = seg ? ns : s_lo;
            for (int sl = sb + tid; sl < se; sl += 256) {
                const float2 x = decim_fetch(P, b, i_first + sl, t_hi, kb0, t_lo);
                const int cb = (sl + 1) / D, o = (sl + 1) - cb * D;
                const int c0 = cb - (o == 0 ? 1 : 0);
                tile[o * PT + (c0 % R) * Wq + c0 / R] = x;
            }
        }
        // fast pairs
        if (k_lo + tid < k_hi) {
            int k = k_lo + tid;
            const int s0 = -a + 2 * k;
            int cb = (s0 + 1) / D, o = (s0 + 1) - cb * D;
            const float4* src = reinterpret_cast<const float4*>(P.in + (size_t)b * P.in_stride + (size_t)((uint64_t)(i_first + s0) - P.n0));
            uint32_t krel = (uint32_t)((uint64_t)(i_first + s0) - P.rot_nbase - ((uint64_t)kb0 << 9));
            const int dO = 512 % D, dC = 512 / D;
            const bool rot = P.rot_enable != 0;
#pragma unroll 4
            for (; k < k_hi; k += 256, src += 256, krel += 512) {
                const float4 v = *src;
                float2 x0 = make_float2(v.x, v.y), x1 = make_float2(v.z, v.w);
                if (rot) {
                    x0 = rot_apply_rel(x0, krel, t_hi, t_lo);
                    x1 = rot_apply_rel(x1, krel + 1, t_hi, t_lo);
                }
                const int c0 = cb - (o == 0 ? 1 : 0);
                tile[o * PT + (c0 % R) * Wq + c0 / R] = x0;
                int o1 = o + 1, c1 = cb;
                if (o1 == D) { o1 = 0; c1 = cb; } else c1 = cb;   // (o1 == 0) => column cb+1-1 = cb
                tile[o1 * PT + (c1 % R) * Wq + c1 / R] = x1;
                o += dO; cb += dC;
                if (o >= D) { o -= D; cb++; }
            }
        }
    }
    __syncthreads();

    // ---- FIR: wave g = phase group g.  Tap (p, j) of output m lives in row (D-p)%D, column m-j-1
    //      (relative to bq0): col = R*lane + cst + t, cst = Jpad-(c+1)*JC = 0 mod R, t = r-jj+JC-1.
    const int g = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int p0 = (g * D) >> 2, p1 = ((g + 1) * D) >> 2;
    const int nchunks = Jpad / JC;
    v2f acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = v2f{0.f, 0.f};
    const v2f* tl = reinterpret_cast<const v2f*>(tile);
    for (int p = p0; p < p1; ++p) {
        const int o = p ? D - p : 0;
        const float* tp = P.taps + p * Jpad;
        for (int c = 0; c < nchunks; ++c) {
            const int cq = (Jpad - (c + 1) * JC) / R;
            const v2f* row = tl + o * PT + cq + lane;
            v2f w[R + JC - 1];
#pragma unroll
            for (int t = 0; t < R + JC - 1; ++t) w[t] = row[(t % R) * Wq + t / R];
#pragma unroll
            for (int jj = 0; jj < JC; ++jj) {
                const float h = tp[c * JC + jj];
                const v2f hv = v2f{h, h};
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] = __builtin_elementwise_fma(hv, w[r - jj + JC - 1], acc[r]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) part[g * TILE + R * lane + r] = make_float2(acc[r].x, acc[r].y);
    __syncthreads();

    if (tid < TILE) {
        const uint64_t m = mt + tid;
        if (m < P.m0 + P.m_count) {
            const float2 r0 = part[tid], r1 = part[TILE + tid], r2 = part[2 * TILE + tid], r3 = part[3 * TILE + tid];
            float2 y;
            y.x = (r0.x + r1.x) + (r2.x + r3.x);
            y.y = (r0.y + r1.y) + (r2.y + r3.y);
            P.out.p[(size_t)b * (P.out.mask + 1u) + ((uint32_t)m & P.out.mask)] = y;
        }
    }
}

static void decim_variant(int variant, int& R, int& JC)
{
    switch (variant) {
    case DECIM_R4_J44: R = 4; JC = 44; break;
    case DECIM_R4_J12: R = 4; JC = 12; break;
    case DECIM_R2_J10: R = 2; JC = 10; break;
    default: R = 1; JC = 14; break;
    }
}
int decim_jc(int variant) { int R, JC; decim_variant(variant, R, JC); return JC; }

size_t decim_lds_bytes(int D, int Jpad, int variant)
{
    int R, JC;
    decim_variant(variant, R, JC);
    const int TILE = 64 * R;
    const int W = TILE + Jpad;
    const int Wq = (W + R - 1) / R;
    const int PT = (R * Wq) | 1;
    return (size_t)(512 + 64 + 4 * TILE + D * PT) * sizeof(float2);
}

void launch_decim(const DecimParams& p, int batch, int variant, hipStream_t s)
{
    int R, JC;
    decim_variant(variant, R, JC);
    const uint32_t tiles = (p.m_count + 64 * R - 1) / (64 * R);
    if (tiles == 0) return;
    DecimParams q = p;
    q.tiles = tiles;
    dim3 grid((tiles + 7) / 8 * 8, batch), block(256);
    const size_t lds = decim_lds_bytes(p.D, p.Jpad, variant);
    switch (variant) {
    case DECIM_R4_J44: hipLaunchKernelGGL((k_decim<4, 44>), grid, block, lds, s, q); break;
    case DECIM_R4_J12: hipLaunchKernelGGL((k_decim<4, 12>), grid, block, lds, s, q); break;
    case DECIM_R2_J10: hipLaunchKernelGGL((k_decim<2, 10>), grid, block, lds, s, q); break;
    default:           hipLaunchKernelGGL((k_decim<1, 14>), grid, block, lds, s, q); break;
    }
}

// ---- history keeper: hist_new[k] = rotated sample at absolute index n0 + n - H + k ----
__global__ __launch_bounds__(256) void k_hist(const HistParams P)
{
    const int b = blockIdx.y;
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= P.hist_len) return;
    const int64_t i = (int64_t)(P.n0 + P.n) - (int64_t)P.hist_len + (int64_t)k;
    float2 x = make_float2(0.f, 0.f);
    if (i >= (int64_t)P.n0) {
        x = P.in[(size_t)b * P.in_stride + (size_t)((uint64_t)i - P.n0)];
        if (P.rot_enable) {
            const uint64_t kk = (uint64_t)i - P.rot_nbase;
            const float2 hi = sincos_turn(P.rot_acc + ((kk >> 9) << 9) * P.rot_inc);
            x = cmul_fma(x, cmul_fma(hi, P.rot_lo[(uint32_t)kk & 511u]));
        }
    } else if (i >= 0) {
        const uint64_t d = P.n0 - (uint64_t)i;  // 1..hist_len
        if (d <= P.hist_len) x = P.hist_old[(size_t)b * P.hist_len + (P.hist_len - (uint32_t)d)];
    }
    P.hist_new[(size_t)b * P.hist_len + k] = x;
}
void launch_hist_save(const HistParams& p, int batch, hipStream_t s)
{
    if (p.hist_len == 0) return;
    dim3 grid((p.hist_len + 255) / 256, batch), block(256);
    hipLaunchKernelGGL(k_hist, grid, block, 0, s, p);
}

// ---- K2: rational resampler I/D.  One fmaf chain per output, j ascending (oracle orc_resamp_ccf).
// Tile of 256 outputs; the input span is staged in LDS once (coalesced), taps [I][Jp] in LDS.
__device__ __forceinline__ float2 resamp_fetch(const ResampParams& P, int b, int64_t i)
{
    if (i < 0) return make_float2(0.f, 0.f);
    const uint64_t ui = (uint64_t)i;
    if (P.in) {
        if (ui >= P.n0 + P.n) return make_float2(0.f, 0.f);
        if (ui >= P.n0) {
            float2 x = P.in[(size_t)b * P.in_stride + (size_t)(ui - P.n0)];
            if (P.rot_enable) {
                const uint64_t kk = ui - P.rot_nbase;
                const float2 hi = sincos_turn(P.rot_acc + ((kk >> 9) << 9) * P.rot_inc);
                x = cmul_fma(x, cmul_fma(hi, P.rot_lo[(uint32_t)kk & 511u]));
            }
            return x;
        }
        const uint64_t d = P.n0 - ui;
        if (d > P.hist_len) return make_float2(0.f, 0.f);
        return P.hist[(size_t)b * P.hist_len + (P.hist_len - (uint32_t)d)];
    }
    if (ui >= P.n0 + P.n) return make_float2(0.f, 0.f);
    return P.in_ring.p[(size_t)b * (P.in_ring.mask + 1u) + ((uint32_t)ui & P.in_ring.mask)];
}

__global__ __launch_bounds__(256) void k_resamp(const ResampParams P, int span)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float* taps = reinterpret_cast<float*>(smem);                  // I * Jp
    float2* xs = reinterpret_cast<float2*>(taps + ((P.I * P.Jp + 3) & ~3));  // span
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const uint32_t T = blockDim.x;                                  // outputs per workgroup: 256, or 64 when the input span of 256 would not leave room for a second workgroup per CU
    const uint64_t q_first = P.q0 + (uint64_t)blockIdx.x * T;
    for (int k = tid; k < P.I * P.Jp; k += (int)T) taps[k] = P.taps[k];
    // input index of output q: c(q) = floor(q*D/I); the tile needs [c(q_first) - (Jp-1), c(q_last)]
    const int64_t c_first = (int64_t)((q_first * (uint64_t)P.D) / (uint64_t)P.I);
    const int64_t base = c_first - (P.Jp - 1);
    for (int k = tid; k < span; k += (int)T) xs[k] = resamp_fetch(P, b, base + k);
    __syncthreads();
    const uint64_t q = q_first + tid;
    if (q >= P.q0 + P.q_count) return;
    const uint64_t u = q * (uint64_t)P.D;
    const int ph = (int)(u % (uint64_t)P.I);
    const int c = (int)((int64_t)(u / (uint64_t)P.I) - base);  // local index of x[c(q)]
    const float* tp = taps + ph * P.Jp;
    float ar = 0.f, ai = 0.f;
    for (int j = 0; j < P.Jp; ++j) {
        const float h = tp[j];
        const float2 x = xs[c - j];
        ar = fmaf(h, x.x, ar);
        ai = fmaf(h, x.y, ai);
    }
    P.out.p[(size_t)b * (P.out.mask + 1u) + ((uint32_t)q & P.out.mask)] = make_float2(ar, ai);
    const uint32_t t = blockIdx.x * T + tid;   // output index inside this call
    if (P.port && t < P.port_cap) P.port[(size_t)b * P.port_cap + t] = make_float2(ar, ai);
    if (t == 0 && P.port_counts) P.port_counts[b * 4 + 0] = P.q_count;
}

void launch_resamp(const ResampParams& p, int batch, hipStream_t s)
{
    if (p.q_count == 0) return;
    // span of inputs for T outputs: ceil((T - 1) * D / I) + 1 + Jp - 1 (+1 slack).  Strongly decimating resamplers (3:125 of gr_demod_dmr /
    // gr_demod_m17: 10 974 inputs = 88 KB for 256 outputs) take 64 outputs per workgroup so that several workgroups share a CU
    int T = 256;
    int span = ((T - 1) * p.D + p.I - 1) / p.I + p.Jp + 2;
    if ((size_t)span * sizeof(float2) > 48 * 1024) { T = 64; span = ((T - 1) * p.D + p.I - 1) / p.I + p.Jp + 2; }
    const size_t lds = (size_t)((p.I * p.Jp + 3) & ~3) * sizeof(float) + (size_t)span * sizeof(float2);
    if (dyn_lds_limit(reinterpret_cast<const void*>(k_resamp), 160 * 1024) != hipSuccess) return;
    dim3 grid((p.q_count + T - 1) / T, batch), block(T);
    hipLaunchKernelGGL(k_resamp, grid, block, lds, s, p, span);
}


// ---- 1:2 decimator + channel FIR in one pass (gr_demod_qpsk / gr_demod_4fsk at 250k / 100k symbols: resampler 1:2 then the RRC or
// low-pass at 500 ksps; reference gr_demod_qpsk.cpp:62-76) ------------------------------------------------------------------------
// Round 2 ran k_decim<4,12> (2.97 ms on C5) and k_fir_ccf_tiled (2.13 ms) with the 500 ksps stream going through HBM in between.  Here
// a workgroup stages 4 176 input samples once, de-interleaved into even / odd samples and, inside each, into 8 IMAGES (sample m of a
// parity lies in image m & 7 at position m >> 3), so that a thread that owns 8 CONSECUTIVE outputs reads its sliding window with
// conflict-free ds_read_b64 (consecutive lanes = consecutive positions).  Per chunk of 8 taps a thread reads 15 samples and does
// 64 complex x real MACs as packed fmas (v_pk_fma_f32: both components in one instruction).  The decimated samples go back into
// the same LDS (again 8 images) and the second filter runs the same way; only its outputs leave the CU.
// Summation order = the oracle's (orc_decim_fir_ccf with G = 4 at D = 2: one chain over the even taps, one over the odd taps, j
// ascending, y = r_even + r_odd; orc_fir_ccf: one chain, k ascending): padding taps are zeros on finite samples, so they add +0.
constexpr int D2_NTH = 256, D2_R = 8;
constexpr int D2_W = 261;                        // positions per image: (40 + 2048) / 8
constexpr int D2_HALO = 24;                      // second-filter taps, padded to a multiple of 8
constexpr int D2_TY = D2_R * D2_NTH - D2_HALO;   // outputs a workgroup delivers
__global__ __launch_bounds__(D2_NTH) void k_dec2_fir(const Dec2FirParams P_)
{
    const Dec2FirParams& P = P_;
    const DecimParams& Q = P.d;
    __shared__ float2 t_lo[512];
    __shared__ float2 t_hi[16];
    __shared__ v2f img[2 * 8 * D2_W];            // [parity][image][position]; reused as [image][position] of the decimated samples
    const int b = blockIdx.y, tid = threadIdx.x;
    const uint64_t y0 = Q.m0 + (uint64_t)blockIdx.x * D2_TY;       // first output of this workgroup
    const int64_t mb = (int64_t)y0 - D2_HALO;                       // thread t owns decimated samples mb + 8 t .. + 7
    const int64_t mq = mb - 40;                                     // image position 0 = decimated index mq .. mq + 7
    // ---- rotator tables (caller's buffer only: carried history is already rotated, engine rings too) ----
    uint32_t kb0 = 0;
    if (Q.rot_enable) {
        t_lo[tid] = Q.rot_lo[tid];
        t_lo[tid + 256] = Q.rot_lo[tid + 256];
        const int64_t i_start = 2 * mq - 1;
        const int64_t first_new = i_start > (int64_t)Q.n0 ? i_start : (int64_t)Q.n0;
        kb0 = (uint32_t)(((uint64_t)first_new - Q.rot_nbase) >> 9);
        if (tid < 16) t_hi[tid] = sincos_turn(Q.rot_acc + ((uint64_t)(kb0 + tid) << 9) * Q.rot_inc);
        __syncthreads();
    }
    // ---- stage: local l = m - mq; even sample x[2 m] and odd sample x[2 m - 1] ----
    // (two passes: every load of the thread is issued before the first one is used -- a fetch -> rotate -> store loop pays the memory
    //  latency nine times per workgroup)
    {
        constexpr int NK = (8 * D2_W + D2_NTH - 1) / D2_NTH;         // 9 pairs per thread
        const float2* inb = Q.in ? Q.in + (size_t)b * Q.in_stride : nullptr;
        const float2* hb = Q.hist ? Q.hist + (size_t)b * Q.hist_len : nullptr;
        const float2* rb = Q.in_ring.p ? Q.in_ring.p + (size_t)b * (Q.in_ring.mask + 1u) : nullptr;
        const int64_t i_end = (int64_t)(Q.n0 + Q.n);
        float2 v[NK][2];
        bool fresh[NK][2];                                           // from the caller's buffer: still to be rotated
        const int64_t i_lo = 2 * mq - 1;
#if defined(QRL_D2_ABL) && QRL_D2_ABL == 2      // developer ablation: no global loads (filters run on whatever the LDS holds)
        if (true) {
#pragma unroll
            for (int k = 0; k < NK; ++k) { v[k][0] = v[k][1] = make_float2((float)tid, 1.0f); fresh[k][0] = fresh[k][1] = false; }
        } else
#endif
        if (inb && i_lo >= (int64_t)Q.n0 && i_lo + 2 * 8 * D2_W <= i_end) {   // the whole span lies in the caller's buffer (workgroup uniform)
            const float2* src = inb + (size_t)(i_lo - (int64_t)Q.n0) + 2 * tid;
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const bool ok = tid + D2_NTH * k < 8 * D2_W;
                v[k][0] = ok ? src[2 * D2_NTH * k] : make_float2(0.f, 0.f);
                v[k][1] = ok ? src[2 * D2_NTH * k + 1] : make_float2(0.f, 0.f);
                fresh[k][0] = fresh[k][1] = Q.rot_enable != 0;
            }
        } else
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int l = tid + D2_NTH * k;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int64_t i = 2 * (mq + l) - 1 + e;              // e = 0: the odd sample x[2 m - 1], e = 1: the even one x[2 m]
                const float2* src = nullptr;
                fresh[k][e] = false;
                if (l < 8 * D2_W && i >= 0 && i < i_end) {
                    if (inb) {
                        if (i >= (int64_t)Q.n0) { src = inb + (size_t)(i - (int64_t)Q.n0); fresh[k][e] = Q.rot_enable != 0; }
                        else if ((int64_t)Q.n0 - i <= (int64_t)Q.hist_len) src = hb + (Q.hist_len - (uint32_t)((int64_t)Q.n0 - i));
                    } else src = rb + ((uint32_t)i & Q.in_ring.mask);
                }
                v[k][e] = src ? *src : make_float2(0.f, 0.f);
            }
        }
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int l = tid + D2_NTH * k;
            if (l < 8 * D2_W) {
                const int64_t i = 2 * (mq + l) - 1;
                float2 xo = v[k][0], xe = v[k][1];
                if (fresh[k][0]) xo = rot_apply(xo, (uint64_t)i - Q.rot_nbase, t_hi, kb0, t_lo);
                if (fresh[k][1]) xe = rot_apply(xe, (uint64_t)(i + 1) - Q.rot_nbase, t_hi, kb0, t_lo);
                const int a = (l & 7) * D2_W + (l >> 3);
                img[a] = v2f{xe.x, xe.y};
                img[8 * D2_W + a] = v2f{xo.x, xo.y};
            }
        }
    }
    __syncthreads();
    // ---- first filter: 8 outputs per thread, two chains each ----
    v2f acc[2][D2_R];
#pragma unroll
    for (int r = 0; r < D2_R; ++r) { acc[0][r] = v2f{0.f, 0.f}; acc[1][r] = v2f{0.f, 0.f}; }
    // a wave whose first decimated sample lies at or behind the end of this call's outputs feeds nobody (the second filter is causal)
    // and stores nothing: it skips both filters.  The last workgroup of a stream is mostly such waves (8 192 outputs = 4 x 2 024 + 96).
    const bool idle_wave = mb + 8 * (int64_t)(tid & ~63) >= (int64_t)(Q.m0 + Q.m_count);
#if defined(QRL_D2_ABL) && QRL_D2_ABL == 1      // developer ablation: no filters (staging, barriers and stores only)
    if (false)
#else
    if (!idle_wave)
#endif
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const v2f* im = img + par * 8 * D2_W + 5 + tid;             // position of decimated index mb + 8 t in image 0
        const float* tp = P.taps + par * 40;
        for (int c = 0; c < 5; ++c) {                                // taps 8 c .. 8 c + 7 of this parity's chain
            v2f w[15];                                               // w[o + 7] = sample at offset o = r - j' from mb + 8 t - 8 c
#pragma unroll
            for (int o = 0; o < 8; ++o) w[7 + o] = im[o * D2_W - c];
#pragma unroll
            for (int o = 1; o < 8; ++o) w[7 - o] = im[(8 - o) * D2_W - c - 1];
            float h[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) h[j] = tp[8 * c + j];
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int r = 0; r < D2_R; ++r)
                    acc[par][r] = __builtin_elementwise_fma(v2f{h[j], h[j]}, w[7 + r - j], acc[par][r]);
        }
    }
    __syncthreads();                                                 // every thread is done with the input images
    v2f* dimg = img;                                                 // decimated sample mb + 8 t + r -> image r, position t
#pragma unroll
    for (int r = 0; r < D2_R; ++r) dimg[r * D2_W + tid] = acc[0][r] + acc[1][r];
    __syncthreads();
    if (blockIdx.x == 0 && tid == 0 && P.counts) P.counts[b * 4 + 0] = Q.m_count;
    const bool live = !(tid < D2_HALO / 8 || idle_wave);             // the first three threads only supplied the halo
    // ---- second filter ----
    v2f y[D2_R];
#pragma unroll
    for (int r = 0; r < D2_R; ++r) y[r] = v2f{0.f, 0.f};
    if (live) {
        const v2f* im = dimg + tid;
        const float* tp = P.taps + 80;
#if defined(QRL_D2_ABL) && QRL_D2_ABL == 1
        for (int c = 0; c < 0; ++c) {
#else
        for (int c = 0; c < D2_HALO / 8; ++c) {
#endif
            v2f w[15];
#pragma unroll
            for (int o = 0; o < 8; ++o) w[7 + o] = im[o * D2_W - c];
#pragma unroll
            for (int o = 1; o < 8; ++o) w[7 - o] = im[(8 - o) * D2_W - c - 1];
            float h[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) h[j] = tp[8 * c + j];
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int r = 0; r < D2_R; ++r)
                    y[r] = __builtin_elementwise_fma(v2f{h[j], h[j]}, w[7 + r - j], y[r]);
        }
    }
    // ---- outputs: the filtered stream (engine ring) and, when asked for, the caller's port-0 buffer.  A thread owns 8 CONSECUTIVE
    // outputs, so a store straight from its registers would scatter 8-byte pieces at a 64-byte stride (every line written eight
    // times by eight instructions).  The outputs go back through the LDS images instead (the decimated samples are dead now) and
    // leave in order: lane = consecutive output, 512 contiguous bytes per store instruction. ----
    v2f* oimg = img + 8 * D2_W;                                      // output mb + 8 t + r -> image r, position t; the upper half (odd input samples) has been dead since the first filter: no barrier against dimg's readers
    if (live)
#pragma unroll
        for (int r = 0; r < D2_R; ++r) oimg[r * D2_W + tid] = y[r];
    __syncthreads();
    const uint64_t m_end = Q.m0 + Q.m_count;
    float2* orow = P.out.p + (size_t)b * (P.out.mask + 1u);
    float2* prow = P.port ? P.port + (size_t)b * P.port_cap : nullptr;
#pragma unroll
    for (int k = 0; k < D2_R; ++k) {
        const int o = tid + D2_NTH * k;                              // output y0 + o = decimated index mb + D2_HALO + o
        const uint64_t m = y0 + (uint64_t)o;
        if (o < D2_TY && m < m_end) {
            const v2f v = oimg[(o & 7) * D2_W + (o >> 3) + D2_HALO / 8];
            const float2 f = make_float2(v.x, v.y);
            orow[(uint32_t)m & P.out.mask] = f;
            const uint64_t t = m - Q.m0;
            if (prow && t < P.port_cap) prow[t] = f;
        }
    }
}
bool dec2_fir_supported(int nt1, int D, int nt2) { return D == 2 && nt1 <= 79 && (nt1 & 1) && nt2 <= D2_HALO; }
uint32_t dec2_fir_lookback() { return 2 * (D2_HALO + 40) + 2; }
// table: [0..39] even taps h[2 j], [40..79] odd taps h[2 j + 1], [80..103] second filter; zero padded
std::vector<float> dec2_fir_table(const std::vector<float>& h1, const std::vector<float>& h2)
{
    std::vector<float> t(80 + D2_HALO, 0.0f);
    for (size_t k = 0; k < h1.size(); ++k) t[(k & 1) * 40 + (k >> 1)] = h1[k];
    for (size_t k = 0; k < h2.size(); ++k) t[80 + k] = h2[k];
    return t;
}
void launch_dec2_fir(const Dec2FirParams& p, int batch, hipStream_t s)
{
    if (p.d.m_count == 0) return;
    hipLaunchKernelGGL(k_dec2_fir, dim3((p.d.m_count + D2_TY - 1) / D2_TY, batch), dim3(D2_NTH), 0, s, p);
}

}  // namespace qrl
