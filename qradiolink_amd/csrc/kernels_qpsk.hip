// kernels_qpsk.hip — the recursive chains with a complex symbol synchroniser, one lane per stream (workgroup = 64 streams):
//   k_qpsk_pipe4    gr_demod_qpsk (reference src/gr/gr_demod_qpsk.cpp:97-123,141-154): agc2_cc(1, 0.1, 1, 1) ->
//                   costas_loop_cc(pi/200/sps, 4, use_snr) -> symbol_sync_cc(MOD_M&M, sps, ..., constellation_dqpsk, MMSE 8 tap) ->
//                   costas_loop_cc(pi/400, 4, use_snr) -> diff_phasor_cc -> multiply_const_cc(e^{-j 3 pi / 4}) -> port 1
//                   (constellation) and the interleaved soft symbols (x48, +128, float_to_uchar) that feed the K=7 Viterbi (k_fec);
//                   every recurrence on its own wave (see the kernel).
//   k_qpsk_loops<1> gr_demod_bpsk (src/gr/gr_demod_bpsk.cpp:54-101): agc2_cc(0.1, 0.1) -> clock_recovery_mm_cc ->
//                   costas_loop_cc(2 pi / 200, 2) -> real part x64 + 128
//   k_qpsk_loops<2> symbol_sync_cc alone on the 4-level rect constellation (gr_demod_4fsk non-FM branch, gr_demod_4fsk.cpp:138-195)
// The blocks are causal per sample, so chaining them inside one serial loop is exact.  k_qpsk_loops: wave 0 runs the recursion out
// of an LDS window, waves 1-3 prefetch the next window of the filtered input (coalesced along the stream) and flush the previous
// window's symbols.  Window k holds samples [k W - 16, (k+1) W): the first 16 columns are carried over from the window before
// (symbol sync looks back at most 13 samples), the rest arrives raw, is overwritten in place by pass 1 (AGC), then pass 2 (clock
// recovery and everything at the symbol rate) walks the row.
#include <cstdlib>
#include "devmath.hpp"
#include "engine.hpp"

namespace qrl {

constexpr int QP_W = 64;
constexpr int QP_BACK = 16;
constexpr int QP_COLS = QP_BACK + QP_W;      // 80
constexpr int QP_PITCH = QP_COLS + 1;        // float2 units, odd
constexpr int QP_OMAX = 38;                  // symbols per stream per window (sps >= 1.9)
constexpr int QP_OPITCH = QP_OMAX + 1;

template <int MODE>
__global__ __launch_bounds__(256) void k_qpsk_loops(const QpskParams P, int batch)
{
    extern __shared__ __align__(16) unsigned char qp_smem[];
    float2* win = reinterpret_cast<float2*>(qp_smem);                    // [2][64][QP_PITCH]
    float2* osym = win + 2 * 64 * QP_PITCH;                              // [2][64][QP_OPITCH]
    float* mm = reinterpret_cast<float*>(osym + 2 * 64 * QP_OPITCH);     // [129][8]
    float* th = mm + 129 * 8;                                            // [256]
    int* ocnt = reinterpret_cast<int*>(th + 256);                        // [2][64]
    uint64_t* obase = reinterpret_cast<uint64_t*>(ocnt + 2 * 64);        // [2][64]
    uint64_t* oo0 = obase + 2 * 64;                                      // [64]

    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int b0 = blockIdx.x * 64;
    const int nstreams = min(64, batch - b0);
    for (int k = tid; k < 129 * 8; k += 256) mm[k] = P.mmse[k];
    th[tid] = P.tanh_tab[tid];

    const uint64_t np0 = P.np0, avail = P.avail;      // samples already through AGC + Costas / available now
    const long long k_first = (long long)(np0 / QP_W);
    const long long k_last = avail > np0 ? (long long)((avail - 1) / QP_W) : k_first - 1;

    QpskState st;
    const bool active = wv == 0 && b0 + lane < batch;
    if (wv == 0) {
        if (active) {
            st = P.st[b0 + lane];
            // Costas outputs [np0 - 16, np0) of the previous call -> their columns of the first window
            float2* row = win + (size_t)(k_first & 1) * 64 * QP_PITCH + lane * QP_PITCH;
            const long long i0 = k_first * QP_W - QP_BACK;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const long long i = (long long)np0 - 16 + j;
                const long long c = i - i0;
                if (c >= 0 && c < QP_COLS) row[c] = st.hist[j];
            }
        } else {
            st = QpskState{};
            st.ii = ~0ull >> 1;
        }
        oo0[lane] = st.oo;
    }

    auto load_window = [&](long long k, int t, int nthreads) {   // raw samples [max(kW, np0), min((k+1)W, avail))
        float2* wbuf = win + (size_t)(k & 1) * 64 * QP_PITCH;
        const long long ia = max((long long)np0, k * QP_W), ib = min((long long)avail, (k + 1) * QP_W);
        const int cnt = (int)(ib - ia);
        if (cnt <= 0) return;
        const int c0 = (int)(ia - (k * QP_W - QP_BACK));
        constexpr int BATCH = 8;
        const int total = nstreams * cnt;
        for (int base = t; base < total; base += nthreads * BATCH) {
            float2 v[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int idx = base + u * nthreads;
                v[u] = make_float2(0.f, 0.f);
                if (idx < total) {
                    const int s = idx / cnt, c = idx - s * cnt;
                    v[u] = P.in.p[(size_t)(b0 + s) * (P.in.mask + 1u) + ((uint32_t)(ia + c) & P.in.mask)];
                }
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int idx = base + u * nthreads;
                if (idx < total) { const int s = idx / cnt, c = idx - s * cnt; wbuf[s * QP_PITCH + c0 + c] = v[u]; }
            }
        }
    };
    auto flush_window = [&](long long k, int t, int nthreads) {
        const int pb = (int)(k & 1);
        const float2* ob = osym + (size_t)pb * 64 * QP_OPITCH;
        for (int idx = t; idx < nstreams * QP_OMAX; idx += nthreads) {
            const int s = idx / QP_OMAX, j = idx - s * QP_OMAX;
            if (j < ocnt[pb * 64 + s]) {
                const float2 v = ob[s * QP_OPITCH + j];
                const uint64_t o = obase[pb * 64 + s] + j;
                if (MODE == 1) {   // complex_to_real -> multiply_const(64) -> add_const(128) -> float_to_uchar (gr_demod_bpsk.cpp:96-101)
                    float q = v.x * P.soft_mul; q = q + P.soft_add;
                    float r = rintf(q);
                    if (!(r >= 0.f)) r = 0.f; if (r > 255.f) r = 255.f;
                    P.soft.p[(size_t)(b0 + s) * (P.soft.mask + 1u) + ((uint32_t)o & P.soft.mask)] = (uint8_t)r;
                    const uint64_t kk1 = o - oo0[s];
                    if (P.port && kk1 < P.port_cap) P.port[(size_t)(b0 + s) * P.port_cap + kk1] = v;
                    continue;
                }
                // MODE 2: complex_to_float -> interleave -> multiply_const(128) -> add_const(128) -> float_to_uchar
                float qa = v.x * P.soft_mul; qa = qa + P.soft_add;
                float qb = v.y * P.soft_mul; qb = qb + P.soft_add;
                float ra = rintf(qa), rb = rintf(qb);
                if (!(ra >= 0.f)) ra = 0.f; if (ra > 255.f) ra = 255.f;
                if (!(rb >= 0.f)) rb = 0.f; if (rb > 255.f) rb = 255.f;
                uint8_t* sp = P.soft.p + (size_t)(b0 + s) * (P.soft.mask + 1u);
                sp[(uint32_t)(2 * o) & P.soft.mask] = (uint8_t)ra;
                sp[(uint32_t)(2 * o + 1) & P.soft.mask] = (uint8_t)rb;
                const uint64_t kk = o - oo0[s];
                if (P.port && kk < P.port_cap) P.port[(size_t)(b0 + s) * P.port_cap + kk] = v;
            }
        }
    };

    __syncthreads();
    if (k_first <= k_last) load_window(k_first, tid, 256);
    __syncthreads();
    for (long long k = k_first; k <= k_last; ++k) {
        if (wv == 0) {
            const int pb = (int)(k & 1);
            float2* row = win + (size_t)pb * 64 * QP_PITCH + lane * QP_PITCH;
            float2* orow = osym + (size_t)pb * 64 * QP_OPITCH + lane * QP_OPITCH;
            const long long i0 = k * QP_W - QP_BACK;
            if (k > k_first && active) {   // carry the last 16 Costas outputs of the previous window
                const float2* prow = win + (size_t)(pb ^ 1) * 64 * QP_PITCH + lane * QP_PITCH;
#pragma unroll
                for (int j = 0; j < QP_BACK; ++j) row[j] = prow[QP_W + j];
            }
            // ---- pass 1: agc2_cc -> costas_loop_cc on the new samples, in place
            const int ca = (int)(max((long long)np0, k * QP_W) - i0), cb = (int)(min((long long)avail, (k + 1) * QP_W) - i0);
            if (active) {
                for (int c = ca; c < cb; ++c) {
                    if (MODE == 2) break;   // symbol_sync_cc consumes the samples as they are
                    // MODE 1: agc2_cc(0.1, 0.1) in place (gr_demod_bpsk.cpp:88-90: no Costas loop in front of the clock recovery)
                    const float2 x = row[c];
                    float2 a; a.x = x.x * st.gain; a.y = x.y * st.gain;
                    const float tmp = -1.0f + sqrtf(a.x * a.x + a.y * a.y);
                    st.gain -= tmp * 0.1f;
                    if (st.gain < 0.0f) st.gain = 10e-5f;
                    if (st.gain > 65536.0f) st.gain = 65536.0f;
                    row[c] = a;
                }
            }
            // ---- pass 2: symbol_sync_cc and the symbol-rate blocks
            const uint64_t wend = (uint64_t)min((long long)avail, (k + 1) * QP_W);   // exclusive
            const uint64_t oo_w = st.oo;
            int nsym = 0;
            while (MODE == 1 && active && st.ii + 8 <= wend && nsym < QP_OMAX) {
                // clock_recovery_mm_cc(sps, 2.5e-5, 0.5, 0.05, 0.001) -> costas_loop_cc(2 pi / 200, 2) (oracle orc_clock_recovery_mm_cc)
                const int off = (int)((long long)st.ii - i0);
                const int imu = (int)rintf(st.mu * 128.0f);
                const float* t = mm + imu * 8;
                float2 y = make_float2(0.f, 0.f);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float2 xs = row[off + j];
                    y.x = fmaf(t[7 - j], xs.x, y.x);
                    y.y = fmaf(t[7 - j], xs.y, y.y);
                }
                st.x2 = st.x1; st.x1 = st.x0; st.x0 = y;
                st.d2 = st.d1; st.d1 = st.d0;
                st.d0.x = y.x > 0.f ? 1.0f : 0.0f; st.d0.y = y.y > 0.f ? 1.0f : 0.0f;     // slicer_0deg
                const float ar = st.d0.x - st.d2.x, ai = st.d0.y - st.d2.y;
                const float xr = ar * st.x1.x + ai * st.x1.y;
                const float br = st.x0.x - st.x2.x, bi = st.x0.y - st.x2.y;
                const float yr = br * st.d1.x + bi * st.d1.y;
                const float mmv = branchless_clip(yr - xr, 1.0f);
                st.avg = st.avg + P.cr_gain_omega * mmv;                                   // d_omega lives in st.avg
                st.avg = P.cr_omega_mid + branchless_clip(st.avg - P.cr_omega_mid, P.cr_omega_lim);
                const float ph = st.mu + st.avg + P.cr_gain_mu * mmv;
                const float fl = floorf(ph);
                st.ii += (uint64_t)(int)fl;
                st.mu = ph - fl;
                const float2 nco = sincos_rad(-st.c2_phase);
                float2 o; o.x = y.x * nco.x - y.y * nco.y; o.y = y.x * nco.y + y.y * nco.x;
                float e2 = o.x * o.y;                                                      // phase_detector_2, use_snr = false
                e2 = branchless_clip(e2, 1.0f);
                st.c2_freq = st.c2_freq + P.c2_beta * e2;
                st.c2_phase = st.c2_phase + st.c2_freq + P.c2_alpha * e2;
                st.c2_phase = phase_wrap(st.c2_phase);
                if (st.c2_freq > 1.0f) st.c2_freq = 1.0f; else if (st.c2_freq < -1.0f) st.c2_freq = -1.0f;
                orow[nsym] = o;
                nsym++;
                st.oo++;
            }
            while (MODE == 2 && active && st.ii + 8 <= wend && nsym < QP_OMAX) {   // symbol_sync_cc alone (gr_demod_4fsk non-FM branch)
                const int off = (int)((long long)st.ii - i0);
                const int imu = (int)rintf(st.mu * 128.0f);
                const float* t = mm + imu * 8;
                float2 y = make_float2(0.f, 0.f);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float2 xs = row[off + j];
                    y.x = fmaf(t[7 - j], xs.x, y.x);
                    y.y = fmaf(t[7 - j], xs.y, y.y);
                }
                st.x2 = st.x1; st.x1 = st.x0; st.x0 = y;
                st.d2 = st.d1; st.d1 = st.d0;
                {   // constellation_rect{-1.5,-0.5,0.5,1.5}: real-axis sector point, imag 0 (oracle slice_real)
                    int sector = (int)((double)y.x + 2.0);
                    sector = sector < 0 ? 0 : (sector > 3 ? 3 : sector);
                    st.d0.x = (float)sector - 1.5f; st.d0.y = 0.0f;
                }
                float e;
                {
                    const float ar = st.x0.x - st.x2.x, ai = st.x0.y - st.x2.y;
                    const float br = st.d0.x - st.d2.x, bi = st.d0.y - st.d2.y;
                    const float u = (ar * st.d1.x + ai * st.d1.y) - (br * st.x1.x + bi * st.x1.y);
                    e = QRL_TED_MODMM_ERROR(QRL_TED_MODMM_CC, u, branchless_clip);   // named contract: include/qrl_contracts.h
                }
                st.avg = st.avg + P.ss_beta * e;
                if (st.avg > P.ss_maxp) st.avg = P.ss_maxp; else if (st.avg < P.ss_minp) st.avg = P.ss_minp;
                st.inst = st.avg + P.ss_alpha * e;
                if (st.inst <= 0.f) st.inst = st.avg;
                const float ph = st.mu + st.inst;
                const float fl = floorf(ph);
                st.mu = ph - fl;
                st.ii += (uint64_t)(int)fl;
                orow[nsym] = y;
                nsym++;
                st.oo++;
            }
            ocnt[pb * 64 + lane] = nsym;
            obase[pb * 64 + lane] = oo_w;
            if (k == k_last && active) {   // keep the last 16 Costas outputs for the next call
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const long long i = (long long)avail - 16 + j;
                    const long long c = i - i0;
                    st.hist[j] = (c >= 0 && i >= 0) ? row[c] : make_float2(0.f, 0.f);
                }
            }
        } else {
            if (k + 1 <= k_last) load_window(k + 1, tid - 64, 192);
            if (k > k_first) flush_window(k - 1, tid - 64, 192);
        }
        __syncthreads();
    }
    if (k_first <= k_last) flush_window(k_last, tid, 256);
    if (active) {
        P.st[b0 + lane] = st;
        P.counts[(b0 + lane) * 4 + 1] = (uint32_t)(st.oo - oo0[lane]);
        if (P.oo_snap) P.oo_snap[b0 + lane] = st.oo;
    }
}

// ---- four-stage wave pipeline of the QPSK chain (MODE 0) -----------------------------------------------------------------------------
// The chain is a serial recurrence per stream; with one lane per stream the time of a call is (samples per stream) x (instructions
// the slowest wave issues per sample), whatever the batch.  So every recurrence gets its own wave (its own SIMD) and nothing but the
// recurrence runs on it:
//   stage 0 agc2_cc                          window t      (in place in the LDS ring)
//   stage 1 costas_loop_cc #1                window t - 1  (in place)
//   stage 2 symbol_sync_cc (MMSE + M&M TED)  window t - 2  -> interpolated symbols Y
//   stage 3 costas_loop_cc #2, diff_phasor, rotate         window t - 3  Y -> osym
//   waves 4-5  load window t + 1 from the RRC ring, flush the symbols of window t - 4 (soft symbols + constellation port)
// one barrier per window.  The samples live in one LDS ring per stream (5 windows of 32 columns + an 8-column mirror of the first
// columns so that the 8-tap interpolator never wraps); the loops are straight-line code (selects instead of branches, inputs of four
// steps loaded ahead of the recurrence).  The arithmetic and its order are those of k_qpsk_loops<0> / the oracle (bit-exact).
#ifndef QRL_Q4_W
#define QRL_Q4_W 32
#endif
// Window = 32 columns: the workgroup holds 137 KB of LDS -- at C5's batch one workgroup on EVERY CU for the kernel's whole 2 ms, so what
// runs beside it must fit 23 KB (k_fec was rebuilt to 0.5 KB per wave for that reason; a k_dec2_fir workgroup, 37 KB, cannot).  A
// 16-column window (77 KB, -DQRL_Q4_W=16) was measured: the per-window overhead makes this kernel 23 % slower alone (2 401 against
// 1 950 us) and C3, which is this kernel's latency, 38 % slower -- rejected (profiles/r04_c5_pipe4_lds_footprint.log).
constexpr int Q4_W = QRL_Q4_W, Q4_NB = 5, Q4_RC = Q4_W * Q4_NB, Q4_MIR = 8;
constexpr int Q4_PITCH = Q4_RC + Q4_MIR + 1;   // 169 float2, odd
constexpr int Q4_OMAX = Q4_W == 32 ? 20 : 11;  // symbols per stream per window (sps >= 1.9: W / 1.9 + 2)
constexpr int Q4_OPITCH = Q4_OMAX + 1;

__device__ __forceinline__ float tanhf_lut_sel(float x, const float* __restrict__ T)   // tanhf_lut without branches
{
    int index = (int)(128.0f + 64.0f * x);
    index = index > 255 ? 255 : (index < 0 ? 0 : index);
    float v = T[index];
    v = x > 2.0f ? 1.0f : v;
    v = x <= -2.0f ? -1.0f : v;
    return v;
}
__device__ __forceinline__ float costas4_snr_error_sel(float2 o, const float* __restrict__ T)
{
    const float snr = (o.x * o.x + o.y * o.y);
    return (tanhf_lut_sel(snr * o.x, T) * o.y) - (tanhf_lut_sel(snr * o.y, T) * o.x);
}
__device__ __forceinline__ float clamp_pm1(float f)   // if (f > 1) f = 1; else if (f < -1) f = -1;
{
    f = f > 1.0f ? 1.0f : f;
    return f < -1.0f ? -1.0f : f;
}
__device__ __forceinline__ int q4_col(long long i) { return (int)(((i % Q4_RC) + Q4_RC) % Q4_RC); }

#ifdef QRL_Q4_PROF
// developer build (tools/kernel_variants.sh kernels_qpsk.hip q4prof -DQRL_Q4_PROF): per wave (= pipeline stage) the ticks between two
// barriers that the wave spent working, and the ticks of the whole loop: which stage sets the pace of the serial walk
__device__ unsigned long long g_q4_prof[6][3];
extern "C" void qrl_q4_prof_read(unsigned long long* out18)
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out18, HIP_SYMBOL(g_q4_prof), sizeof(unsigned long long) * 18);
    static unsigned long long zero[18];
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_q4_prof), zero, sizeof zero);
}
#endif
__global__ __launch_bounds__(384) void k_qpsk_pipe4(const QpskParams P, int batch)
{
    extern __shared__ __align__(16) unsigned char qp_smem[];
    float2* ring = reinterpret_cast<float2*>(qp_smem);                   // [64][Q4_PITCH]
    float2* ysym = ring + 64 * Q4_PITCH;                                 // [2][64][Q4_OPITCH]
    float2* osym = ysym + 2 * 64 * Q4_OPITCH;                            // [2][64][Q4_OPITCH]
    float* mm = reinterpret_cast<float*>(osym + 2 * 64 * Q4_OPITCH);     // [129][8]
    float* th = mm + 129 * 8;                                            // [256]
    int* ycnt = reinterpret_cast<int*>(th + 256);                        // [2][64]
    int* ocnt = ycnt + 2 * 64;                                           // [2][64]
    uint64_t* obase = reinterpret_cast<uint64_t*>(ocnt + 2 * 64);        // [2][64]
    uint64_t* oo0 = obase + 2 * 64;                                      // [64]

    const int tid = threadIdx.x, hwv = tid >> 6, lane = tid & 63;
    // stage of this wave.  Consecutive waves of a workgroup go to consecutive SIMDs, so the two loader waves (4, 5) share theirs with waves
    // 0 and 1: those are the AGC and the symbol synchroniser (45 % / 56 % busy), the two Costas loops -- the first one sets the pace of the
    // walk -- have a SIMD each (wave 1 <-> wave 2: k_qpsk_pipe4 2 020-2 050 -> 1 970-1 980 us on C3, profiles/r06_c5_experiments.log)
    const int wv = hwv == 1 ? 2 : hwv == 2 ? 1 : hwv;
    const int b0 = blockIdx.x * 64;
    const int nstreams = min(64, batch - b0);
#ifndef QRL_Q4_NOPRIO
    // a serial walk whose run time IS the receiver's step (C3) shares its SIMDs with throughput kernels of the neighbouring calls: its
    // few instructions go first
    __builtin_amdgcn_s_setprio(3);
#endif
    for (int k = tid; k < 129 * 8; k += 384) mm[k] = P.mmse[k];
    if (tid < 256) th[tid] = P.tanh_tab[tid];
    const uint64_t np0 = P.np0, avail = P.avail;
    const long long k_first = (long long)(np0 / Q4_W);
    const long long k_last = avail > np0 ? (long long)((avail - 1) / Q4_W) : k_first - 1;
    const bool active = b0 + lane < batch;
    QpskState* gst = P.st + (b0 + (active ? lane : 0));
    float2* row = ring + lane * Q4_PITCH;

    // per-wave recurrence state
    float gain = 0.f, c1_phase = 0.f, c1_freq = 0.f;
    uint64_t ii = ~0ull >> 1, oo = 0;
    float mu = 0.f, avg = 0.f, inst = 0.f;
    float2 x0 = {0.f, 0.f}, x1 = x0, x2 = x0, d0 = x0, d1 = x0, d2 = x0, dprev = x0;
    float c2_phase = 0.f, c2_freq = 0.f;
    if (active) {
        if (wv == 0) gain = gst->gain;
        if (wv == 1) {
            c1_phase = gst->c1_phase; c1_freq = gst->c1_freq;
#pragma unroll
            for (int j = 0; j < 16; ++j) {   // Costas outputs [np0 - 16, np0) of the previous call
                const int c = q4_col((long long)np0 - 16 + j);
                const float2 v = gst->hist[j];
                row[c] = v;
                if (c < Q4_MIR) row[Q4_RC + c] = v;
            }
        }
        if (wv == 2) {
            ii = gst->ii; mu = gst->mu; avg = gst->avg; inst = gst->inst;
            x0 = gst->x0; x1 = gst->x1; x2 = gst->x2; d0 = gst->d0; d1 = gst->d1; d2 = gst->d2;
        }
        if (wv == 3) { c2_phase = gst->c2_phase; c2_freq = gst->c2_freq; dprev = gst->dprev; oo = gst->oo; oo0[lane] = oo; }
    } else if (wv == 3) {
        oo0[lane] = 0;
    }
    if (k_first > k_last) {   // nothing new
        if (wv == 3 && active) { P.counts[(b0 + lane) * 4 + 1] = 0; if (P.oo_snap) P.oo_snap[b0 + lane] = oo; }
        return;
    }

    auto load_window = [&](long long k, int t, int nthreads) {   // raw samples [max(kW, np0), min((k+1)W, avail))
        const long long ia = max((long long)np0, k * Q4_W), ib = min((long long)avail, (k + 1) * Q4_W);
        const int cnt = (int)(ib - ia);
        if (cnt <= 0) return;
        const int c0 = (int)(k % Q4_NB) * Q4_W + (int)(ia - k * Q4_W);
        const int total = nstreams * cnt;
        constexpr int BATCH = 8;
        for (int base = t; base < total; base += nthreads * BATCH) {
            float2 v[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int idx = base + u * nthreads;
                v[u] = make_float2(0.f, 0.f);
                if (idx < total) {
                    const int s = cnt == Q4_W ? idx / Q4_W : idx / cnt, c = idx - s * cnt;
                    v[u] = P.in.p[(size_t)(b0 + s) * (P.in.mask + 1u) + ((uint32_t)(ia + c) & P.in.mask)];
                }
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int idx = base + u * nthreads;
                if (idx < total) { const int s = cnt == Q4_W ? idx / Q4_W : idx / cnt, c = idx - s * cnt; ring[s * Q4_PITCH + c0 + c] = v[u]; }
            }
        }
    };
    auto flush_window = [&](long long k, int t, int nthreads) {
        const int pb = (int)(k & 1);
        const float2* ob = osym + (size_t)pb * 64 * Q4_OPITCH;
        for (int idx = t; idx < nstreams * Q4_OMAX; idx += nthreads) {
            const int s = idx / Q4_OMAX, j = idx - s * Q4_OMAX;
            if (j < ocnt[pb * 64 + s]) {
                const float2 v = ob[s * Q4_OPITCH + j];
                const uint64_t o = obase[pb * 64 + s] + j;
                // complex_to_float -> interleave -> multiply_const(48) -> add_const(128) -> float_to_uchar
                float qa = v.x * P.soft_mul; qa = qa + P.soft_add;
                float qb = v.y * P.soft_mul; qb = qb + P.soft_add;
                float ra = rintf(qa), rb = rintf(qb);
                if (!(ra >= 0.f)) ra = 0.f; if (ra > 255.f) ra = 255.f;
                if (!(rb >= 0.f)) rb = 0.f; if (rb > 255.f) rb = 255.f;
                uint8_t* sp = P.soft.p + (size_t)(b0 + s) * (P.soft.mask + 1u);
                sp[(uint32_t)(2 * o) & P.soft.mask] = (uint8_t)ra;
                sp[(uint32_t)(2 * o + 1) & P.soft.mask] = (uint8_t)rb;
                const uint64_t kk = o - oo0[s];
                if (P.port && kk < P.port_cap) P.port[(size_t)(b0 + s) * P.port_cap + kk] = v;
            }
        }
    };
    auto agc_step = [&](float2 x) {
        float2 a; a.x = x.x * gain; a.y = x.y * gain;
        const float tmp = -1.0f + sqrtf(a.x * a.x + a.y * a.y);
        const float rate = tmp > gain ? 1.0f : 0.1f;          // agc2_cc(attack 1, decay 0.1)
        gain -= tmp * rate;
        gain = gain < 0.0f ? 10e-5f : gain;
        gain = gain > 65536.0f ? 65536.0f : gain;
        return a;
    };
    auto costas1_step = [&](float2 a) {
        const float2 nco = sincos_rad(-c1_phase);             // (cos, sin)
        float2 o; o.x = a.x * nco.x - a.y * nco.y; o.y = a.x * nco.y + a.y * nco.x;
        float e = costas4_snr_error_sel(o, th);
        e = branchless_clip(e, 1.0f);
        c1_freq = c1_freq + P.c1_beta * e;
        c1_phase = c1_phase + c1_freq + P.c1_alpha * e;
        c1_phase = phase_wrap(c1_phase);
        c1_freq = clamp_pm1(c1_freq);
        return o;
    };

    __syncthreads();
    load_window(k_first, tid, 384);
    __syncthreads();
    const float SQ = 0.707107f;
#ifdef QRL_Q4_PROF
    unsigned long long busy = 0, total0 = __builtin_readcyclecounter();
#endif
    for (long long t = k_first; t <= k_last + 4; ++t) {
#ifdef QRL_Q4_PROF
        const unsigned long long tb0 = __builtin_readcyclecounter();
#endif
        if (wv == 0) {
            if (t <= k_last && active) {                      // ---- agc2_cc on window t, in place
                const int base = (int)(t % Q4_NB) * Q4_W;
                const int ca = base + (int)(max((long long)np0, t * Q4_W) - t * Q4_W), cb = base + (int)(min((long long)avail, (t + 1) * Q4_W) - t * Q4_W);
                int c = ca;
                for (; c + 4 <= cb; c += 4) {
                    float2 x[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) x[u] = row[c + u];
#pragma unroll
                    for (int u = 0; u < 4; ++u) x[u] = agc_step(x[u]);
#pragma unroll
                    for (int u = 0; u < 4; ++u) row[c + u] = x[u];
                }
                for (; c < cb; ++c) row[c] = agc_step(row[c]);
            }
        } else if (wv == 1) {
            const long long k = t - 1;
            if (k >= k_first && k <= k_last && active) {      // ---- first Costas loop on window k, in place
                const int base = (int)(k % Q4_NB) * Q4_W;
                const int ca = base + (int)(max((long long)np0, k * Q4_W) - k * Q4_W), cb = base + (int)(min((long long)avail, (k + 1) * Q4_W) - k * Q4_W);
                int c = ca;
                for (; c + 4 <= cb; c += 4) {
                    float2 x[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) x[u] = row[c + u];
#pragma unroll
                    for (int u = 0; u < 4; ++u) x[u] = costas1_step(x[u]);
#pragma unroll
                    for (int u = 0; u < 4; ++u) row[c + u] = x[u];
                }
                for (; c < cb; ++c) row[c] = costas1_step(row[c]);
                if (base == 0) {
#pragma unroll
                    for (int j = 0; j < Q4_MIR; ++j) row[Q4_RC + j] = row[j];
                }
            }
        } else if (wv == 2) {
            const long long k = t - 2;
            if (k >= k_first && k <= k_last) {                // ---- symbol_sync_cc over window k
                const int pb = (int)(k & 1);
                float2* yrow = ysym + (size_t)pb * 64 * Q4_OPITCH + lane * Q4_OPITCH;
                const uint64_t wend = (uint64_t)min((long long)avail, (k + 1) * Q4_W);
                int nsym = 0;
                int off = active ? (int)(ii % Q4_RC) : 0;
                while (active && ii + 8 <= wend && nsym < Q4_OMAX) {
                    const int imu = (int)rintf(mu * 128.0f);
                    const float* tp = mm + imu * 8;
                    float2 y = make_float2(0.f, 0.f);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float2 xs = row[off + j];
                        y.x = fmaf(tp[7 - j], xs.x, y.x);
                        y.y = fmaf(tp[7 - j], xs.y, y.y);
                    }
                    x2 = x1; x1 = x0; x0 = y;
                    d2 = d1; d1 = d0;
                    d0.x = y.x > 0.f ? SQ : -SQ; d0.y = y.y > 0.f ? SQ : -SQ;
                    const float ar = x0.x - x2.x, ai = x0.y - x2.y;
                    const float br = d0.x - d2.x, bi = d0.y - d2.y;
                    const float u = (ar * d1.x + ai * d1.y) - (br * x1.x + bi * x1.y);
                    const float e = QRL_TED_MODMM_ERROR(QRL_TED_MODMM_CC, u, branchless_clip);   // named contract: include/qrl_contracts.h
                    avg = avg + P.ss_beta * e;
                    avg = avg > P.ss_maxp ? P.ss_maxp : (avg < P.ss_minp ? P.ss_minp : avg);
                    inst = avg + P.ss_alpha * e;
                    inst = inst <= 0.f ? avg : inst;
                    const float ph = mu + inst;
                    const float fl = floorf(ph);
                    mu = ph - fl;
                    const int adv = (int)fl;
                    ii += (uint64_t)adv;
                    off += adv;
                    off = off >= Q4_RC ? off - Q4_RC : off;
                    yrow[nsym] = y;
                    nsym++;
                }
                ycnt[pb * 64 + lane] = nsym;
            }
        } else if (wv == 3) {
            const long long k = t - 3;
            if (k >= k_first && k <= k_last) {                // ---- second Costas loop, diff_phasor, rotation on the symbols of window k
                const int pb = (int)(k & 1);
                const float2* yrow = ysym + (size_t)pb * 64 * Q4_OPITCH + lane * Q4_OPITCH;
                float2* orow = osym + (size_t)pb * 64 * Q4_OPITCH + lane * Q4_OPITCH;
                const int n = ycnt[pb * 64 + lane];
                obase[pb * 64 + lane] = oo;
                ocnt[pb * 64 + lane] = n;
                for (int j = 0; j < n; ++j) {
                    const float2 y = yrow[j];
                    const float2 nco = sincos_rad(-c2_phase);
                    float2 o; o.x = y.x * nco.x - y.y * nco.y; o.y = y.x * nco.y + y.y * nco.x;
                    float e2 = costas4_snr_error_sel(o, th);
                    e2 = branchless_clip(e2, 1.0f);
                    c2_freq = c2_freq + P.c2_beta * e2;
                    c2_phase = c2_phase + c2_freq + P.c2_alpha * e2;
                    c2_phase = phase_wrap(c2_phase);
                    c2_freq = clamp_pm1(c2_freq);
                    float2 dp; dp.x = o.x * dprev.x + o.y * dprev.y; dp.y = o.y * dprev.x - o.x * dprev.y;
                    dprev = o;
                    float2 v; v.x = dp.x * P.rot.x - dp.y * P.rot.y; v.y = dp.x * P.rot.y + dp.y * P.rot.x;
                    orow[j] = v;
                }
                oo += (uint64_t)n;
            }
        } else {
            if (t + 1 <= k_last) load_window(t + 1, tid - 256, 128);
            if (t - 4 >= k_first && t - 4 <= k_last) flush_window(t - 4, tid - 256, 128);
        }
#ifdef QRL_Q4_PROF
        busy += __builtin_readcyclecounter() - tb0;
#endif
        __syncthreads();
    }
#ifdef QRL_Q4_PROF
    if (lane == 0) { atomicAdd(&g_q4_prof[wv][0], busy); atomicAdd(&g_q4_prof[wv][1], __builtin_readcyclecounter() - total0); atomicAdd(&g_q4_prof[wv][2], 1ull); }
#endif
    if (!active) return;
    if (wv == 0) gst->gain = gain;
    if (wv == 1) {
        gst->c1_phase = c1_phase; gst->c1_freq = c1_freq;
#pragma unroll
        for (int j = 0; j < 16; ++j) {     // the last 16 Costas outputs for the next call
            const long long i = (long long)avail - 16 + j;
            gst->hist[j] = i >= 0 ? row[q4_col(i)] : make_float2(0.f, 0.f);
        }
    }
    if (wv == 2) {
        gst->ii = ii; gst->mu = mu; gst->avg = avg; gst->inst = inst;
        gst->x0 = x0; gst->x1 = x1; gst->x2 = x2; gst->d0 = d0; gst->d1 = d1; gst->d2 = d2;
    }
    if (wv == 3) {
        gst->c2_phase = c2_phase; gst->c2_freq = c2_freq; gst->dprev = dprev; gst->oo = oo;
        P.counts[(b0 + lane) * 4 + 1] = (uint32_t)(oo - oo0[lane]);
        if (P.oo_snap) P.oo_snap[b0 + lane] = oo;
    }
}
static size_t qpsk_pipe4_lds_bytes()
{
    return (size_t)(64 * Q4_PITCH + 4 * 64 * Q4_OPITCH) * sizeof(float2) + (129 * 8 + 256) * sizeof(float) + 4 * 64 * sizeof(int) +
           (2 * 64 + 64) * sizeof(uint64_t);
}

static size_t qpsk_lds_bytes()
{
    return (size_t)(2 * 64 * QP_PITCH + 2 * 64 * QP_OPITCH) * sizeof(float2) + (129 * 8 + 256) * sizeof(float) + 2 * 64 * sizeof(int) +
           (2 * 64 + 64) * sizeof(uint64_t);
}

void launch_qpsk_loops(const QpskParams& p, int batch, hipStream_t s)
{
    if (p.mode == 0) {   // gr_demod_qpsk chain: four-stage wave pipeline
        if (dyn_lds_limit(reinterpret_cast<const void*>(k_qpsk_pipe4), (int)qpsk_pipe4_lds_bytes()) != hipSuccess) return;
        hipLaunchKernelGGL(k_qpsk_pipe4, dim3((batch + 63) / 64), dim3(384), qpsk_pipe4_lds_bytes(), s, p, batch);
        return;
    }
    const void* kp = p.mode == 2 ? reinterpret_cast<const void*>(k_qpsk_loops<2>) : reinterpret_cast<const void*>(k_qpsk_loops<1>);
    if (dyn_lds_limit(kp, (int)qpsk_lds_bytes()) != hipSuccess) return;
    if (p.mode == 2) hipLaunchKernelGGL(k_qpsk_loops<2>, dim3((batch + 63) / 64), dim3(256), qpsk_lds_bytes(), s, p, batch);
    else hipLaunchKernelGGL(k_qpsk_loops<1>, dim3((batch + 63) / 64), dim3(256), qpsk_lds_bytes(), s, p, batch);
}

}  // namespace qrl
