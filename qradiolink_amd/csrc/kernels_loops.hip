// kernels_loops.hip — the recursive (sample-serial) blocks of the RX chains, ONE LANE PER STREAM.
// These loops cannot be scanned exactly (sincos / slicer decisions sit inside the recurrence), so
// parallelism comes from the batch: a wave advances 64 independent streams in lock step.  Input is
// staged per wave through LDS in coalesced rows ([stream][window], odd pitch => conflict-free
// column walks) so that the serial part never waits on HBM/L2 latency per item.
//   k_fll         fll_band_edge_cc            (gr_demod_2fsk.cpp:90)
//   k_symsync_ff  symbol_sync_ff + soft-symbol quantiser (gr_demod_2fsk.cpp:106-118,
//                 gr_demod_gmsk.cpp:89-101): TED (mod-)M&M, MMSE 8-tap interpolator, PI clock loop
#include "devmath.hpp"
#include "engine.hpp"

namespace qrl {

// ------------------------------------------------------------------ FLL band edge
// fll_band_edge_cc: y[n] = x[n - NT] * nco(phase);  u/l = band-edge FIRs over the last NT outputs y;
// error = |l|^2 - |u|^2;  2nd-order loop.  Summation contract (oracle orc_fll_band_edge): each of the four
// accumulators is ONE fmaf chain over the taps OLDEST SAMPLE FIRST, so only the last link of the chain
// depends on the sample that was just derotated.  (A variant with the delay line in absolute-index register
// slots and an NT-times unrolled body was tried: 20 % slower -- its 60 KB of code thrashes the instruction
// cache and a single wave per SIMD is issue-latency bound anyway: ~5 cycles per instruction.)
// Geometry: 4 waves per workgroup (64 streams x 4 lanes, one wave per SIMD) and a 128-sample LDS window.  (Single-wave
// workgroups with a 16-sample window and <= 96 registers were tried so that the kernel could slip in beside the front end of
// the next call in overlapped mode: the recursion itself got slower -- 2.5 instead of 1.7 ms -- and the overlap no better.)
// Two geometries (template parameters FLL_TH threads per workgroup, FLL_CH samples per stream and LDS window, powers of two):
//   256 x 32   the default form: 4 waves per workgroup (64 streams x 4 lanes, one wave per SIMD), 17 KB of LDS (68 KB with the 128-sample
//              window of rounds 2-3);
//   64 x 16    the SLIM form of the overlapped mode: single-wave workgroups with 3 KB of LDS and <= 216 VGPRs, which the dispatcher can
//              place on a CU whose LDS and wave slots are otherwise full of front-end workgroups (k_decim_pm leaves 16 KB of LDS and
//              224 VGPRs per SIMD): the recursion of call k then runs UNDER the front end of call k + 1 instead of behind it.

#ifndef QRL_FLL_CH
#define QRL_FLL_CH 32    // samples per stream and LDS window of the 4-wave geometry (64 streams x QRL_FLL_CH x 8 bytes of LDS).  128 until round 4;
                         // with the symbol synchroniser no longer starved (r04) the 17 KB window is worth 1.1 % of a C1 step, same-box A/B x 2:
                         // 8.15 against 8.24 ms (16 samples: 8.24).  Results do not depend on it.
#endif
#ifndef QRL_FLL_TH
#define QRL_FLL_TH 256   // threads per workgroup of the default geometry (4 lanes per stream): 256 = one wave per SIMD and CU at 16 k streams
#endif
template <int CTRL> __device__ __forceinline__ float dpp_quad(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// FOUR LANES PER STREAM.  The two NT-tap band-edge filters are the bulk of the arithmetic but only their newest link sits on
// the recursion's critical path, so the delay line is cut into 4 groups of NT/4 samples, one per lane of a quad: every lane
// runs the short oldest-first fmaf chains of its group, the delay line shifts through the quad with one DPP move, and the
// partial sums meet by two DPP butterflies as (p0 + p1) + (p2 + p3) -- the summation contract of oracle orc_fll_band_edge.
// The NCO / loop update is computed redundantly by the four lanes (bit-identical inputs, bit-identical results).  Compared with
// the lane-per-stream kernel there are 4x more waves (one per SIMD instead of one per CU at 16k streams) with ~3x shorter
// instruction streams.
template <int NT, int FLL_TH, int FLL_CH>
__global__ __launch_bounds__(FLL_TH) void k_fll(const FllParams P, int batch)
{
    constexpr int FLL_NS = FLL_TH / 4;       // streams per workgroup
    constexpr int GL = NT / 4;
    __shared__ float2 win[FLL_NS][FLL_CH + 1];
    __shared__ float2 tl[NT], tu[NT];
    __shared__ float2 dump[FLL_TH];
    const int tid = threadIdx.x;
    const int sl = tid >> 2, g = tid & 3;
    const int b0 = blockIdx.x * FLL_NS;
    const int b = b0 + sl;
    const bool active = b < batch;
    for (int k = tid; k < NT; k += FLL_TH) { tl[k] = P.lower[k]; tu[k] = P.upper[k]; }
    float phase = 0.f, freq = 0.f;
    float2 dl[GL];   // dl[t] = y[n - (g GL + t)]
    if (active) {
        const FllState& s = P.st[b];
        phase = s.phase; freq = s.freq;
#pragma unroll
        for (int t = 0; t < GL; ++t) dl[t] = s.dl[g * GL + t];
    } else {
#pragma unroll
        for (int t = 0; t < GL; ++t) dl[t] = make_float2(0.f, 0.f);
    }
    const int nstreams = min(FLL_NS, batch - b0);
    __syncthreads();
    float2 hu[GL], hl[GL];   // this lane's taps: entry j of the device tables belongs to y[n - j]
#pragma unroll
    for (int t = 0; t < GL; ++t) { hu[t] = tu[g * GL + t]; hl[t] = tl[g * GL + t]; }
    // Staging: every thread keeps the NEXT window's items in registers (NPT unconditional 8-byte loads issued before the serial
    // loop of the current window, so their latency hides behind it; a load-wait-store loop here cost as much as the recursion).
    constexpr int NPT = FLL_NS * FLL_CH / FLL_TH;
    float2 pre[NPT];
    auto preload = [&](uint32_t c0) {
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int idx = tid + FLL_TH * i, s = min(idx / FLL_CH, nstreams - 1), k = idx % FLL_CH;
            const int64_t a = (int64_t)(P.q0 + c0 + k) - NT;   // x[n - NT]; ring reads are in bounds for any index
            const float2 v = P.in.p[(size_t)(b0 + s) * (P.in.mask + 1u) + ((uint32_t)a & P.in.mask)];
            pre[i] = a >= 0 ? v : make_float2(0.f, 0.f);
        }
    };
    preload(0);
    for (uint32_t c0 = 0; c0 < P.count; c0 += FLL_CH) {
        const int len = min((uint32_t)FLL_CH, P.count - c0);
        __syncthreads();   // the flush of the previous window is through with win
#pragma unroll
        for (int i = 0; i < NPT; ++i) { const int idx = tid + FLL_TH * i; win[idx / FLL_CH][idx % FLL_CH] = pre[i]; }
        __syncthreads();
        if (c0 + FLL_CH < P.count) preload(c0 + FLL_CH);
        if (active) {
            for (int k = 0; k < len; ++k) {
                const float2 x = win[sl][k];
                // critical path: NCO -> derotated sample -> newest link of lane 0 -> butterflies -> loop update.  The body is ONE
                // basic block (no lane-dependent branch: lanes 1-3 of a quad send their copy of y to a dump slot), so the scheduler
                // can fill the dependency gaps of this chain with the independent work below.
                const float2 nco = sincos_rad(phase);  // (cos, sin)
                // independent of this sample's NCO: shift the delay line through the quad -- lane g takes the oldest entry of lane
                // g - 1 -- and run the oldest-first chains over everything but the newest slot
                float2 carry;
                carry.x = dpp_quad<0x90>(dl[GL - 1].x);
                carry.y = dpp_quad<0x90>(dl[GL - 1].y);
#pragma unroll
                for (int t = GL - 1; t > 0; --t) dl[t] = dl[t - 1];
                float ur = 0.f, ui = 0.f, lr = 0.f, li = 0.f;
#pragma unroll
                for (int t = GL - 1; t >= 1; --t) {
                    const float2 v = dl[t];
                    ur = fmaf(hu[t].x, v.x, ur); ur = fmaf(-hu[t].y, v.y, ur);
                    ui = fmaf(hu[t].x, v.y, ui); ui = fmaf(hu[t].y, v.x, ui);
                    lr = fmaf(hl[t].x, v.x, lr); lr = fmaf(-hl[t].y, v.y, lr);
                    li = fmaf(hl[t].x, v.y, li); li = fmaf(hl[t].y, v.x, li);
                }
                const float2 y = cmul(x, nco);
                *(g == 0 ? &win[sl][k] : &dump[tid]) = y;   // output staged in place (lane 0 of the quad), flushed coalesced below
                dl[0] = g == 0 ? y : carry;
                {
                    const float2 v = dl[0];
                    ur = fmaf(hu[0].x, v.x, ur); ur = fmaf(-hu[0].y, v.y, ur);
                    ui = fmaf(hu[0].x, v.y, ui); ui = fmaf(hu[0].y, v.x, ui);
                    lr = fmaf(hl[0].x, v.x, lr); lr = fmaf(-hl[0].y, v.y, lr);
                    li = fmaf(hl[0].x, v.y, li); li = fmaf(hl[0].y, v.x, li);
                }
                // (p0 + p1) + (p2 + p3): quad butterflies (lane ^ 1, then lane ^ 2); float addition commutes, so all four
                // lanes end up with the same bits
                ur = ur + dpp_quad<0xB1>(ur); ui = ui + dpp_quad<0xB1>(ui); lr = lr + dpp_quad<0xB1>(lr); li = li + dpp_quad<0xB1>(li);
                ur = ur + dpp_quad<0x4E>(ur); ui = ui + dpp_quad<0x4E>(ui); lr = lr + dpp_quad<0x4E>(lr); li = li + dpp_quad<0x4E>(li);
                const float error = (lr * lr + li * li) - (ur * ur + ui * ui);
                freq = freq + P.beta * error;
                phase = phase + freq + P.alpha * error;
                phase = phase_wrap(phase);
                freq = __builtin_fminf(__builtin_fmaxf(freq, -P.max_freq), P.max_freq);   // == the two-sided clamp for finite freq
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int idx = tid + FLL_TH * i, s = idx / FLL_CH, k = idx % FLL_CH;
            if (s < nstreams && k < len) P.out.p[(size_t)(b0 + s) * (P.out.mask + 1u) + ((uint32_t)(P.q0 + c0 + k) & P.out.mask)] = win[s][k];
        }
    }
    if (active) {
        FllState& s = P.st[b];
        if (g == 0) { s.phase = phase; s.freq = freq; }
#pragma unroll
        for (int t = 0; t < GL; ++t) s.dl[g * GL + t] = dl[t];
    }
}

void launch_fll(const FllParams& p, int batch, hipStream_t s)
{
    if (!p.count) return;
    if (p.slim) {
        dim3 grid((batch + 15) / 16), block(64);
        if (p.nt == 16) hipLaunchKernelGGL((k_fll<16, 64, 16>), grid, block, 0, s, p, batch);
        else            hipLaunchKernelGGL((k_fll<32, 64, 16>), grid, block, 0, s, p, batch);
        return;
    }
    constexpr int TH = QRL_FLL_TH;
    dim3 grid((batch + TH / 4 - 1) / (TH / 4)), block(TH);
    if (p.nt == 16) hipLaunchKernelGGL((k_fll<16, TH, QRL_FLL_CH>), grid, block, 0, s, p, batch);
    else            hipLaunchKernelGGL((k_fll<32, TH, QRL_FLL_CH>), grid, block, 0, s, p, batch);
}

// ------------------------------------------------------------------ symbol_sync_ff
// One workgroup = SS_NS = 32 streams.  Wave 0 runs the recursion, one lane per stream, out of an LDS window; waves
// 1-3 meanwhile fetch the NEXT window of all 64 streams (coalesced along the stream) and flush the symbols of
// the PREVIOUS window, so the serial wave never waits on L2/HBM latency (that wait was > 50 % of the old
// single-wave kernel).  Windows sit on an absolute grid: window k holds samples [k W - SS_BACK, (k+1) W + 8)
// of every stream; a lane works while its 8-tap interpolator fits, then all lanes move on together (cursors of
// different streams never drift apart by more than a symbol inside a call: every lane consumes all samples).
// Geometry <NS streams per workgroup, W new samples per window>: <32, 192> everywhere (half a wave of streams keeps the LDS under 80 KB,
// so that the kernel fits beside ONE front-end workgroup and overlaps the next call); <16, 96> = 25 KB for the multi-carrier receiver,
// whose symbol synchroniser has to slip in beside the four 39 KB workgroups per CU of the NEXT call's per-channel kernel (a 75 KB
// workgroup waits until two of those retire on the same CU at once: 0.56 of its 1.38 ms stayed exposed per C4 step).
template <int NS, int W> struct SsGeo {
    static constexpr int BACK = 16;                   // samples kept in front of the grid point
    static constexpr int COLS = BACK + W + 8;
    static constexpr int PITCH = COLS + 1;            // odd pitch: lanes walk down columns conflict-free
    static constexpr int OMAX = W == 192 ? 56 : (W + 5) * 2 / 7 + 1;     // symbols per stream per window: room for (W + 5) / 3.5 (56 at W = 192, 29 at W = 96)
    static constexpr int OPITCH = OMAX + 1;
    static constexpr size_t lds_bytes() { return (size_t)(2 * NS * PITCH + 4 + 129 * 8 + 2 * NS * OPITCH) * sizeof(float) + 2 * 64 * sizeof(int) + (2 * 64 + 64 + 2) * sizeof(uint64_t); }
};

__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint64_t o = __shfl_xor((unsigned long long)v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}

template <int SS_NS, int SS_W, bool VEC = false, int SS_TH = 256>
__global__ __launch_bounds__(SS_TH) void k_symsync_ff(const SymSyncParams P, int batch)
{
    using G = SsGeo<SS_NS, SS_W>;
    constexpr int SS_BACK = G::BACK, SS_COLS = G::COLS, SS_PITCH = G::PITCH, SS_OMAX = G::OMAX, SS_OPITCH = G::OPITCH;
    extern __shared__ __align__(16) unsigned char ss_smem[];
    float* win = reinterpret_cast<float*>(ss_smem);              // [2][SS_NS][SS_PITCH]
    float* mm = win + 2 * SS_NS * SS_PITCH + 4;                         // [129][8]
    float* osym = mm + 129 * 8;                                  // [2][SS_NS][SS_OPITCH]
    int* ocnt = reinterpret_cast<int*>(osym + 2 * SS_NS * SS_OPITCH);       // [2][64]
    uint64_t* obase = reinterpret_cast<uint64_t*>(ocnt + 2 * 64);        // [2][64]
    uint64_t* oo0 = obase + 2 * 64;                                      // [64]
    long long* kfl = reinterpret_cast<long long*>(oo0 + 64);             // [2]: first / last window

    const int tid = threadIdx.x;
    const int wv = tid >> 6, lane = tid & 63;
    const int b0 = blockIdx.x * SS_NS;
    const int nstreams = min(SS_NS, batch - b0);
    for (int k = tid; k < 129 * 8; k += SS_TH) mm[k] = P.mmse[k];

    SymSyncState st;
    bool active = false;
    if (wv == 0) {
        active = lane < SS_NS && b0 + lane < batch;
        if (active) st = P.st[b0 + lane];
        else { st.ii = ~0ull >> 1; st.oo = 0; st.mu = 0; st.avg = st.inst = 0; st.x0 = st.x1 = st.x2 = st.d0 = st.d1 = st.d2 = 0; }
        if (lane < SS_NS) oo0[lane] = st.oo;
        const uint64_t lo = wave_min_u64(active ? st.ii : ~0ull);
        if (lane == 0) {
            // windows that can hold a symbol: ii + 8 <= avail
            kfl[0] = (long long)(lo / SS_W);
            kfl[1] = P.avail >= 8 ? (long long)((P.avail - 8) / SS_W) : -1;
            if (lo + 8 > P.avail) kfl[1] = kfl[0] - 1;
        }
    }
    __syncthreads();
    const long long k_first = kfl[0], k_last = kfl[1];

    // loader: window k of all streams -> win[k & 1]
    auto load_window_scalar = [&](long long k, int t, int nthreads) {
        float* wbuf = win + (size_t)(k & 1) * SS_NS * SS_PITCH;
        const long long i0 = k * SS_W - SS_BACK;
        constexpr int BATCH = 12;
        const int total = nstreams * SS_COLS;
        for (int base = t; base < total; base += nthreads * BATCH) {
            float v[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int idx = base + u * nthreads;
                v[u] = 0.f;
                if (idx < total) {
                    const int s = idx / SS_COLS, c = idx - s * SS_COLS;
                    const long long i = i0 + c;
                    if (i >= 0 && (uint64_t)i < P.avail)
                        v[u] = P.in.p[(size_t)(b0 + s) * (P.in.mask + 1u) + ((uint32_t)i & P.in.mask)];
                }
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int idx = base + u * nthreads;
                if (idx < total) { const int s = idx / SS_COLS, c = idx - s * SS_COLS; wbuf[s * SS_PITCH + c] = v[u]; }
            }
        }
    };
    // the same with four consecutive samples per load (VEC; the multi-carrier receiver's geometry, round 5): the window starts at a multiple of 16 samples and the ring rows are powers of
    // two, so a group of four never wraps and is 16-byte aligned (a quarter of the load instructions of the scalar form, whose serialised round trips -- not the
    // bytes -- were what a loader wave spent its time on); groups that touch the stream's start or its end take the checked scalar path.
    auto load_window_vec = [&](long long k, int t, int nthreads) {
        float* wbuf = win + (size_t)(k & 1) * SS_NS * SS_PITCH;
        const long long i0 = k * SS_W - SS_BACK;
        static_assert(SS_COLS % 4 == 0 && SS_W % 16 == 0 && SS_BACK % 16 == 0, "window geometry: groups of four");
        constexpr int C4 = SS_COLS / 4;
        // all of a helper thread's groups in flight at once (one round trip per window instead of several), 16 at most
        constexpr int PER = (SS_NS * C4 + (SS_TH - 64) - 1) / (SS_TH - 64);
        constexpr int BATCH = PER < 16 ? PER : 16;
        const int total = nstreams * C4;
        for (int base = t; base < total; base += nthreads * BATCH) {
            float4 v[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int idx = base + u * nthreads;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (idx < total) {
                    const int s = idx / C4, c = (idx - s * C4) * 4;
                    const long long i = i0 + c;
                    const float* row = P.in.p + (size_t)(b0 + s) * (P.in.mask + 1u);
                    if (i >= 0 && (uint64_t)(i + 3) < P.avail) v[u] = *reinterpret_cast<const float4*>(row + ((uint32_t)i & P.in.mask));
                    else {
                        if (i >= 0 && (uint64_t)i < P.avail) v[u].x = row[(uint32_t)i & P.in.mask];
                        if (i + 1 >= 0 && (uint64_t)(i + 1) < P.avail) v[u].y = row[(uint32_t)(i + 1) & P.in.mask];
                        if (i + 2 >= 0 && (uint64_t)(i + 2) < P.avail) v[u].z = row[(uint32_t)(i + 2) & P.in.mask];
                        if (i + 3 >= 0 && (uint64_t)(i + 3) < P.avail) v[u].w = row[(uint32_t)(i + 3) & P.in.mask];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int idx = base + u * nthreads;
                if (idx < total) {
                    const int s = idx / C4, c = (idx - s * C4) * 4;
                    float* w = wbuf + s * SS_PITCH + c;
                    w[0] = v[u].x; w[1] = v[u].y; w[2] = v[u].z; w[3] = v[u].w;
                }
            }
        }
    };
    auto load_window = [&](long long k, int t, int nthreads) { if constexpr (VEC) load_window_vec(k, t, nthreads); else load_window_scalar(k, t, nthreads); };
    // flusher: symbols of window k -> soft-symbol ring (multiply_const -> add_const -> float_to_uchar) and port 1
    auto flush_window = [&](long long k, int t, int nthreads) {
        const int pb = (int)(k & 1);
        const float* ob = osym + (size_t)pb * SS_NS * SS_OPITCH;
        for (int idx = t; idx < nstreams * SS_OMAX; idx += nthreads) {
            const int s = idx / SS_OMAX, j = idx - s * SS_OMAX;
            if (j < ocnt[pb * 64 + s]) {
                const float y = ob[s * SS_OPITCH + j];
                if (P.tail == 1) {   // gr_demod_dmr.cpp:73-105 (x0.9), gr_demod_m17.cpp:74-101 (x1): -> phase_modulator_fc(pi/2) -> slicer -> pack -> map{3,1,2,0} -> unpack
                    const uint64_t o = obase[pb * 64 + s] + j;
                    const uint64_t kk = o - oo0[s];
                    const float2 cs = sincos_rad(1.57079632679489661923f * (y * P.tail_scale));
                    if (P.port && kk < P.port_cap) P.port[(size_t)(b0 + s) * P.port_cap + kk] = cs;
                    const int v = ((cs.x >= 0.0f) ? 2 : 0) | ((cs.y >= 0.0f) ? 1 : 0);
                    const int m = (0x27 >> (2 * v)) & 3;   // map {3,1,2,0}
                    if (P.bits && 2 * kk + 1 < P.bits_cap) {
                        P.bits[(size_t)(b0 + s) * P.bits_cap + 2 * kk] = (uint8_t)((m >> 1) & 1);
                        P.bits[(size_t)(b0 + s) * P.bits_cap + 2 * kk + 1] = (uint8_t)(m & 1);
                    }
                    continue;
                }
                if (P.tail == 2) {   // gr_demod_4fsk.cpp:140-146,165-195 (FM): phase_modulator_fc(pi/2) -> port 1; (imag, real) interleaved soft symbols
                    const uint64_t o = obase[pb * 64 + s] + j;
                    const uint64_t kk = o - oo0[s];
                    const float2 cs = sincos_rad(1.57079632679489661923f * y);
                    if (P.port && kk < P.port_cap) P.port[(size_t)(b0 + s) * P.port_cap + kk] = cs;
                    float qa = cs.y * P.soft_mul; qa = qa + P.soft_add;
                    float qb = cs.x * P.soft_mul; qb = qb + P.soft_add;
                    float ra = rintf(qa), rb = rintf(qb);
                    if (!(ra >= 0.f)) ra = 0.f; if (ra > 255.f) ra = 255.f;
                    if (!(rb >= 0.f)) rb = 0.f; if (rb > 255.f) rb = 255.f;
                    uint8_t* sp = P.soft.p + (size_t)(b0 + s) * (P.soft.mask + 1u);
                    sp[(uint32_t)(2 * o) & P.soft.mask] = (uint8_t)ra;
                    sp[(uint32_t)(2 * o + 1) & P.soft.mask] = (uint8_t)rb;
                    continue;
                }
                float q = y * P.soft_mul;
                q = q + P.soft_add;
                float r = rintf(q);
                if (!(r >= 0.f)) r = 0.f;
                if (r > 255.f) r = 255.f;
                const uint64_t o = obase[pb * 64 + s] + j;
                P.soft.p[(size_t)(b0 + s) * (P.soft.mask + 1u) + ((uint32_t)o & P.soft.mask)] = (uint8_t)r;
                const uint64_t kk = o - oo0[s];
                if (P.port && kk < P.port_cap) P.port[(size_t)(b0 + s) * P.port_cap + kk] = make_float2(y, 0.f);
            }
        }
    };

    if (k_first <= k_last) load_window(k_first, tid, SS_TH);
    __syncthreads();
    for (long long k = k_first; k <= k_last; ++k) {
        if (wv == 0) {
            const int pb = (int)(k & 1);
            const int ln = lane & (SS_NS - 1);
            const float* row = win + (size_t)pb * SS_NS * SS_PITCH + ln * SS_PITCH;
            float* orow = osym + (size_t)pb * SS_NS * SS_OPITCH + ln * SS_OPITCH;
            const long long i0 = k * SS_W - SS_BACK;
            const uint64_t wend = min((uint64_t)((k + 1) * SS_W + 8), P.avail);   // exclusive
            const uint64_t oo_w = st.oo;
            int nsym = 0;
            // in-window cursor as a plain int: the 64-bit absolute cursor is rebuilt after the loop
            int off = active ? (int)min((long long)st.ii - i0, (long long)(1 << 20)) : (1 << 20);
            const int off_end = (int)((long long)wend - i0) - 8;   // last admissible cursor
            while (off <= off_end && nsym < SS_OMAX) {
                const int imu = (int)rintf(st.mu * 128.0f);
                const float4 ta = *reinterpret_cast<const float4*>(mm + imu * 8);
                const float4 tb = *reinterpret_cast<const float4*>(mm + imu * 8 + 4);
                float y = 0.f;
                y = fmaf(tb.w, row[off + 0], y); y = fmaf(tb.z, row[off + 1], y);
                y = fmaf(tb.y, row[off + 2], y); y = fmaf(tb.x, row[off + 3], y);
                y = fmaf(ta.w, row[off + 4], y); y = fmaf(ta.z, row[off + 5], y);
                y = fmaf(ta.y, row[off + 6], y); y = fmaf(ta.x, row[off + 7], y);
                st.x2 = st.x1; st.x1 = st.x0; st.x0 = y;
                st.d2 = st.d1; st.d1 = st.d0;
                if (P.slicer == 0) st.d0 = (y > 0.f) ? 1.0f : -1.0f;
                else {   // constellation_rect{-1.5,-0.5,0.5,1.5}: sector (int)(re / 1.0 + 4 / 2.0) clamped to [0, 3] (oracle slice_real)
                    int sector = (int)((double)y + 2.0);
                    sector = sector < 0 ? 0 : (sector > 3 ? 3 : sector);
                    st.d0 = (float)sector - 1.5f;
                }
                float e;
                if (P.ted == 0) e = st.d1 * st.x0 - st.d0 * st.x1;
                else {
                    const float u = ((st.x0 - st.x2) * st.d1) - ((st.d0 - st.d2) * st.x1);
                    e = QRL_TED_MODMM_ERROR(QRL_TED_MODMM_FF, u, branchless_clip);   // named contract: include/qrl_contracts.h
                }
                st.avg = st.avg + P.beta * e;
                if (st.avg > P.maxp) st.avg = P.maxp; else if (st.avg < P.minp) st.avg = P.minp;
                st.inst = st.avg + P.alpha * e;
                if (st.inst <= 0.f) st.inst = st.avg;
                const float ph = st.mu + st.inst;
                const float fl = floorf(ph);
                st.mu = ph - fl;
                orow[nsym] = y;
                nsym++;
                off += (int)fl;
            }
            if (active) { st.ii = (uint64_t)(i0 + off); st.oo += (uint64_t)nsym; }
            if (lane < SS_NS) { ocnt[pb * 64 + lane] = nsym; obase[pb * 64 + lane] = oo_w; }
        } else {
            if (k + 1 <= k_last) load_window(k + 1, tid - 64, SS_TH - 64);
            if (k > k_first) flush_window(k - 1, tid - 64, SS_TH - 64);
        }
        __syncthreads();
    }
    if (k_first <= k_last) flush_window(k_last, tid, SS_TH);
    if (wv == 0 && active) {
        P.st[b0 + lane] = st;
        P.counts[(b0 + lane) * 4 + 1] = (uint32_t)(st.oo - oo0[lane]);
        if (P.tail == 1) P.counts[(b0 + lane) * 4 + 2] = 2u * (uint32_t)(st.oo - oo0[lane]);
    }
}

size_t symsync_lds_bytes() { return SsGeo<32, 192>::lds_bytes(); }

#ifndef QRL_CHAN_SS_TH
#define QRL_CHAN_SS_TH 256   // threads per workgroup of the multi-carrier receiver's geometry: one recursion wave + (QRL_CHAN_SS_TH / 64 - 1) loader / flusher waves
#endif
#ifndef QRL_CHAN_SS_NS
#define QRL_CHAN_SS_NS 32    // round 5: <32 streams, 96 samples> = 128 workgroups of 45 KB for the 4096 channel streams of C4 instead of 256 of 25 KB: the recursion is as
#define QRL_CHAN_SS_W 96     // fast alone (1.75 against 1.68 ms) and costs the per-channel kernel beside it less: 2.82 - 2.83 against 2.86 - 2.94 ms per step, four
#endif                       // alternating passes (profiles/r05_c4_experiments.log); 64 streams per wave make the recursion 1.8 x slower (bank conflicts)
void launch_symsync_ff(const SymSyncParams& p, int batch, hipStream_t s)
{
    if (p.slim == 2 && (QRL_CHAN_SS_NS != 16 || QRL_CHAN_SS_W != 96)) {   // the multi-carrier receiver's own geometry (QRL_CHAN_SS_NS streams per workgroup, QRL_CHAN_SS_W samples per window)
        const auto k = k_symsync_ff<QRL_CHAN_SS_NS, QRL_CHAN_SS_W, true, QRL_CHAN_SS_TH>;
        const size_t lds = SsGeo<QRL_CHAN_SS_NS, QRL_CHAN_SS_W>::lds_bytes();
        if (dyn_lds_limit(reinterpret_cast<const void*>(k), (int)lds) != hipSuccess) return;
        hipLaunchKernelGGL(k, dim3((batch + QRL_CHAN_SS_NS - 1) / QRL_CHAN_SS_NS), dim3(QRL_CHAN_SS_TH), lds, s, p, batch);
        return;
    }
    if (p.slim) {   // the multi-carrier receiver's geometry (see SsGeo)
        const auto k = k_symsync_ff<16, 96>;
        const size_t lds = SsGeo<16, 96>::lds_bytes();
        if (dyn_lds_limit(reinterpret_cast<const void*>(k), (int)lds) != hipSuccess) return;
        hipLaunchKernelGGL(k, dim3((batch + 15) / 16), dim3(256), lds, s, p, batch);
        return;
    }
    const auto k = k_symsync_ff<32, 192>;
    if (dyn_lds_limit(reinterpret_cast<const void*>(k), (int)symsync_lds_bytes()) != hipSuccess) return;
    hipLaunchKernelGGL(k, dim3((batch + 31) / 32), dim3(256), symsync_lds_bytes(), s, p, batch);
}

}  // namespace qrl
