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
constexpr int FLL_CH = 96;   // samples per stream per LDS window

template <int NT>
__global__ __launch_bounds__(64) void k_fll(const FllParams P, int batch)
{
    __shared__ float2 win[64][FLL_CH + 1];
    __shared__ float2 tl[NT], tu[NT];
    const int lane = threadIdx.x;
    const int b0 = blockIdx.x * 64;
    const int b = b0 + lane;
    const bool active = b < batch;
    if (lane < NT) { tl[lane] = P.lower[lane]; tu[lane] = P.upper[lane]; }
    float phase = 0.f, freq = 0.f;
    float2 dl[NT];
    if (active) {
        const FllState& s = P.st[b];
        phase = s.phase; freq = s.freq;
#pragma unroll
        for (int j = 0; j < NT; ++j) dl[j] = s.dl[j];
    } else {
#pragma unroll
        for (int j = 0; j < NT; ++j) dl[j] = make_float2(0.f, 0.f);
    }
    const int nstreams = min(64, batch - b0);
    constexpr int KPS = (FLL_CH + 63) / 64;
    for (uint32_t c0 = 0; c0 < P.count; c0 += FLL_CH) {
        const int len = min((uint32_t)FLL_CH, P.count - c0);
        __syncthreads();
        // stage x[n - NT] for n in [q0+c0, q0+c0+len) of every stream of this wave; 8 streams of loads in flight
        for (int s0 = 0; s0 < nstreams; s0 += 8) {
            float2 v[8][KPS];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int kk = 0; kk < KPS; ++kk) {
                    const int k = lane + 64 * kk;
                    const int64_t i = (int64_t)(P.q0 + c0 + k) - NT;
                    v[u][kk] = make_float2(0.f, 0.f);
                    if (s0 + u < nstreams && k < len && i >= 0)
                        v[u][kk] = P.in.p[(size_t)(b0 + s0 + u) * (P.in.mask + 1u) + ((uint32_t)i & P.in.mask)];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int kk = 0; kk < KPS; ++kk) {
                    const int k = lane + 64 * kk;
                    if (s0 + u < nstreams && k < FLL_CH) win[s0 + u][k] = v[u][kk];
                }
            }
        }
        __syncthreads();
        if (active) {
            for (int k = 0; k < len; ++k) {
                const float2 x = win[lane][k];
                const float2 nco = sincos_rad(phase);  // (cos, sin)
                const float2 y = cmul(x, nco);
                win[lane][k] = y;                      // output staged in place, flushed coalesced below
#pragma unroll
                for (int j = NT - 1; j > 0; --j) dl[j] = dl[j - 1];
                dl[0] = y;
                float ur = 0.f, ui = 0.f, lr = 0.f, li = 0.f;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const float2 hu = tu[j], hl = tl[j], v = dl[j];
                    ur = fmaf(hu.x, v.x, ur); ur = fmaf(-hu.y, v.y, ur);
                    ui = fmaf(hu.x, v.y, ui); ui = fmaf(hu.y, v.x, ui);
                    lr = fmaf(hl.x, v.x, lr); lr = fmaf(-hl.y, v.y, lr);
                    li = fmaf(hl.x, v.y, li); li = fmaf(hl.y, v.x, li);
                }
                const float error = (lr * lr + li * li) - (ur * ur + ui * ui);
                freq = freq + P.beta * error;
                phase = phase + freq + P.alpha * error;
                phase = phase_wrap(phase);
                if (freq > P.max_freq) freq = P.max_freq; else if (freq < -P.max_freq) freq = -P.max_freq;
            }
        }
        __syncthreads();
        for (int s = 0; s < nstreams; ++s) {
            float2* orow = P.out.p + (size_t)(b0 + s) * (P.out.mask + 1u);
#pragma unroll
            for (int kk = 0; kk < KPS; ++kk) {
                const int k = lane + 64 * kk;
                if (k < len) orow[(uint32_t)(P.q0 + c0 + k) & P.out.mask] = win[s][k];
            }
        }
    }
    if (active) {
        FllState& s = P.st[b];
        s.phase = phase; s.freq = freq;
#pragma unroll
        for (int j = 0; j < NT; ++j) s.dl[j] = dl[j];
    }
}

void launch_fll(const FllParams& p, int batch, hipStream_t s)
{
    if (!p.count) return;
    dim3 grid((batch + 63) / 64), block(64);
    if (p.nt == 16) hipLaunchKernelGGL((k_fll<16>), grid, block, 0, s, p, batch);
    else            hipLaunchKernelGGL((k_fll<32>), grid, block, 0, s, p, batch);
}

// ------------------------------------------------------------------ symbol_sync_ff
constexpr int SS_WIN = 184;          // new samples per window
constexpr int SS_LEN = SS_WIN + 8;   // + interpolator span
constexpr int SS_PITCH = SS_LEN + 1; // odd
constexpr int SS_OPITCH = 65;        // symbols produced per stream per window < 64 (sps >= 3.9)

__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint64_t o = __shfl_xor((unsigned long long)v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}

__global__ __launch_bounds__(64) void k_symsync_ff(const SymSyncParams P, int batch)
{
    __shared__ float win[64 * SS_PITCH];
    __shared__ float mm[129 * 9];
    __shared__ float osym[64 * SS_OPITCH];
    __shared__ int ocnt[64];
    __shared__ uint64_t obase[64], oo0[64];
    const int lane = threadIdx.x;
    const int b0 = blockIdx.x * 64;
    const int b = b0 + lane;
    const bool active = b < batch;
    for (int k = lane; k < 129 * 8; k += 64) mm[(k >> 3) * 9 + (k & 7)] = P.mmse[k];
    SymSyncState st;
    if (active) st = P.st[b];
    else { st.ii = ~0ull >> 1; st.oo = 0; st.mu = 0; st.avg = st.inst = 0; st.x0 = st.x1 = st.x2 = st.d0 = st.d1 = st.d2 = 0; }
    const uint64_t oo_start = st.oo;
    oo0[lane] = oo_start;
    const int nstreams = min(64, batch - b0);
    const float* row = win + lane * SS_PITCH;
    constexpr int KPS = SS_LEN / 64;
    static_assert(SS_LEN % 64 == 0, "window must be a multiple of the wave size");
    while (true) {
        const bool can = active && (st.ii + 8 <= P.avail);
        if (!__any(can)) break;
        const uint64_t w0 = wave_min_u64(can ? st.ii : ~0ull);
        const uint64_t wend = (w0 + SS_LEN < P.avail) ? (w0 + SS_LEN) : P.avail;  // exclusive
        __syncthreads();
        for (int s0 = 0; s0 < nstreams; s0 += 8) {   // 8 streams x KPS coalesced loads in flight
            float v[8][KPS];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int kk = 0; kk < KPS; ++kk) {
                    const uint64_t i = w0 + lane + 64 * kk;
                    v[u][kk] = 0.f;
                    if (s0 + u < nstreams && i < P.avail)
                        v[u][kk] = P.in.p[(size_t)(b0 + s0 + u) * (P.in.mask + 1u) + ((uint32_t)i & P.in.mask)];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int kk = 0; kk < KPS; ++kk)
                    if (s0 + u < nstreams) win[(s0 + u) * SS_PITCH + lane + 64 * kk] = v[u][kk];
            }
        }
        __syncthreads();
        const uint64_t oo_w = st.oo;   // first symbol index of this window for this stream
        int nsym = 0;
        if (can) {
            while (st.ii + 8 <= wend && nsym < 64) {
                const int off = (int)(st.ii - w0);
                const int imu = (int)rintf(st.mu * 128.0f);
                const float* t = mm + imu * 9;
                float y = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) y = fmaf(t[7 - k], row[off + k], y);
                st.x2 = st.x1; st.x1 = st.x0; st.x0 = y;
                st.d2 = st.d1; st.d1 = st.d0; st.d0 = (y > 0.f) ? 1.0f : -1.0f;
                float e;
                if (P.ted == 0) e = st.d1 * st.x0 - st.d0 * st.x1;
                else {
                    const float u = ((st.x0 - st.x2) * st.d1) - ((st.d0 - st.d2) * st.x1);
                    e = branchless_clip(u / 2.0f, 1.0f);
                }
                st.avg = st.avg + P.beta * e;
                if (st.avg > P.maxp) st.avg = P.maxp; else if (st.avg < P.minp) st.avg = P.minp;
                st.inst = st.avg + P.alpha * e;
                if (st.inst <= 0.f) st.inst = st.avg;
                const float ph = st.mu + st.inst;
                const float fl = floorf(ph);
                st.mu = ph - fl;
                osym[lane * SS_OPITCH + nsym] = y;
                nsym++;
                st.oo++;
                st.ii += (uint64_t)(int)fl;
            }
        }
        ocnt[lane] = nsym;
        obase[lane] = oo_w;
        __syncthreads();
        // coalesced flush: soft symbols (multiply_const -> add_const -> float_to_uchar) and port 1
        for (int s = 0; s < nstreams; ++s) {
            const int cnt = ocnt[s];
            if (lane < cnt) {
                const float y = osym[s * SS_OPITCH + lane];
                float v = y * P.soft_mul;
                v = v + P.soft_add;
                float r = rintf(v);
                if (!(r >= 0.f)) r = 0.f;
                if (r > 255.f) r = 255.f;
                const uint64_t o = obase[s] + lane;
                P.soft.p[(size_t)(b0 + s) * (P.soft.mask + 1u) + ((uint32_t)o & P.soft.mask)] = (uint8_t)r;
                const uint64_t k = o - oo0[s];
                if (P.port && k < P.port_cap) P.port[(size_t)(b0 + s) * P.port_cap + k] = make_float2(y, 0.f);
            }
        }
    }
    if (active) {
        P.st[b] = st;
        P.counts[b * 4 + 1] = (uint32_t)(st.oo - oo_start);
    }
}

void launch_symsync_ff(const SymSyncParams& p, int batch, hipStream_t s)
{
    dim3 grid((batch + 63) / 64), block(64);
    hipLaunchKernelGGL(k_symsync_ff, grid, block, 0, s, p, batch);
}

}  // namespace qrl
