// kernels_analog.hip — the analogue voice receivers behind their channel filter:
//   gr_demod_nbfm  reference src/gr/gr_demod_nbfm.cpp:31-88   squelch -> quadrature demod -> 2:5 -> audio filter -> de-emphasis -> x2
//   gr_demod_am    reference src/gr/gr_demod_am.cpp:28-79     squelch -> |.| -> agc2_ff -> DC block -> x0.99 -> 2:5 -> audio filter
//   gr_demod_wbfm  reference src/gr/gr_demod_wbfm.cpp:28-72   squelch -> quadrature demod -> x0.9 -> de-emphasis -> 1:25
// The squelch GATES (pwr_squelch_cc(-140, 0.01, ramp, true)): a muted item is dropped, so everything behind it runs on a
// per-stream item count that only the device knows.  k_an_gate (lane per stream, serial: power estimate, state machine and the
// recursions that sit directly behind it) compacts the unmuted items into a float ring and keeps the cumulative count; the
// feed-forward kernels behind it (k_an_resamp, k_an_fir: thread per output) derive their output range from that count, the last
// recursion (NBFM de-emphasis) is again a lane per stream.  Rates: 20 ksps / 200 ksps in, 8 ksps out.
// Arithmetic = oracle/orc_analog.c, bit for bit (double-precision single-pole filters are plain IEEE mul / add, no contraction).
#include "devmath.hpp"
#include "engine.hpp"

namespace qrl {

__device__ __forceinline__ float2 an_ringc_at(const RingC& r, int b, int64_t i)
{
    if (i < 0) return make_float2(0.f, 0.f);
    return r.p[(size_t)b * (r.mask + 1u) + ((uint32_t)i & r.mask)];
}
__device__ __forceinline__ float an_ringf_at(const RingF& r, int b, int64_t i)
{
    if (i < 0) return 0.f;
    return r.p[(size_t)b * (r.mask + 1u) + ((uint32_t)i & r.mask)];
}
// items the stage behind the gate has produced once g items passed it: rational resampler I:D, or (I == 0) the cessb stretcher,
// which emits whole chunks of 1024 and reads two items ahead
__device__ __forceinline__ uint64_t an_decim_count(uint64_t n, int I, int D)
{
    if (I == 0) return n >= 2 ? 1024 * ((n - 2) / 1024) : 0;
    return n ? ((n - 1) * (uint64_t)I + (uint64_t)I - 1) / (uint64_t)D + 1 : 0;
}

// fft_filter_ccc as the direct FIR it implements (AM channel filter, 571 complex taps at 20 ksps): one fmaf chain per component,
// k ascending, re += hr xr; re += (-hi) xi; im += hr xi; im += hi xr
__global__ __launch_bounds__(256) void k_an_fir_ccc(const FirCccParams P)
{
    extern __shared__ float2 an_taps[];
    for (int k = threadIdx.x; k < P.nt; k += 256) an_taps[k] = P.taps[k];
    __syncthreads();
    const int b = blockIdx.y;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= P.count) return;
    if (t == 0 && P.counts) P.counts[b * 4 + 0] = P.count;
    const int64_t n = (int64_t)(P.q0 + t);
    float ar = 0.f, ai = 0.f;
    const int kmax = n + 1 < (int64_t)P.nt ? (int)(n + 1) : P.nt;
    for (int k = 0; k < kmax; ++k) {
        const float2 h = an_taps[k];
        const float2 x = an_ringc_at(P.in, b, n - k);
        ar = fmaf(h.x, x.x, ar);
        ar = fmaf(-h.y, x.y, ar);
        ai = fmaf(h.x, x.y, ai);
        ai = fmaf(h.y, x.x, ai);
    }
    const float2 y = make_float2(ar, ai);
    P.out.p[(size_t)b * (P.out.mask + 1u) + ((uint32_t)n & P.out.mask)] = y;
    if (P.port && t < P.port_cap) P.port[(size_t)b * P.port_cap + t] = y;
}
void launch_an_fir_ccc(const FirCccParams& p, int batch, hipStream_t s)
{
    if (!p.count) return;
    hipLaunchKernelGGL(k_an_fir_ccc, dim3((p.count + 255) / 256, batch), dim3(256), (size_t)p.nt * sizeof(float2), s, p);
}

// KIND 0 NBFM, 1 AM, 2 WBFM, 3 SSB (gr_demod_ssb.cpp:28-81: squelch -> agc2_cc -> cessb clipper, complex items out)
template <int KIND>
__global__ __launch_bounds__(64) void k_an_gate(const AnGateParams P, int batch)
{
    __shared__ float T[257];
    __shared__ float env_tab[AN_MAX_RAMP + 1];
    __shared__ float2 xq[16][64];
    for (int k = threadIdx.x; k < 257; k += 64) T[k] = P.atan_tab[k];
    for (int k = threadIdx.x; k <= P.ramp; k += 64) env_tab[k] = P.env[k];
    __syncthreads();
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= batch) return;
    AnState st = P.st[b];
    st.g_prev = st.g;
    float* out = P.out.p + (size_t)b * (P.out.mask + 1u);
    // the loop is a chain of dependent operations per item; the loads are not: sixteen items are fetched ahead of the recursion
    constexpr int AN_PF = 16;
    for (uint32_t t0 = 0; t0 < P.count; t0 += AN_PF) {
#pragma unroll
    for (int k = 0; k < AN_PF; ++k) xq[k][threadIdx.x] = an_ringc_at(P.in, b, (int64_t)(P.q0 + t0 + k));   // (ring reads are in bounds for any index; a lane only reads back its own column)
    for (int k = 0; k < AN_PF; ++k) {
        if (t0 + k >= P.count) break;
        const float2 x = xq[k][threadIdx.x];
        const float p = x.x * x.x + x.y * x.y;
        st.pwr = P.alpha * (double)p + P.one_minus_alpha * st.pwr;
        const bool mute = st.pwr < P.threshold;
        switch (st.state) {
        case 0: if (!mute) st.state = P.ramp ? 1 : 2; break;
        case 2: if (mute) st.state = P.ramp ? 3 : 0; break;
        case 1:
            st.env = env_tab[++st.ramped];
            if (st.ramped >= P.ramp) { st.state = 2; st.env = 1.0f; }
            break;
        case 3:
            st.env = env_tab[--st.ramped];
            if (st.ramped == 0) st.state = 0;
            break;
        }
        if (st.state == 0) continue;   // gated
        float2 s;
        s.x = x.x * st.env - x.y * 0.0f;
        s.y = x.x * 0.0f + x.y * st.env;
        if (KIND == 3) {
            float2 o = make_float2(s.x * st.gain, s.y * st.gain);                // agc2_cc(0.1, 0.1, 0.25, 1)
            const float tmp = -P.ref + sqrtf(o.x * o.x + o.y * o.y);
            float rate = P.decay;
            if (tmp > st.gain) rate = P.attack;
            st.gain -= tmp * rate;
            if (st.gain < 0.0f) st.gain = 10e-5f;
            if (st.gain > 65536.0f) st.gain = 65536.0f;
            const float mag = sqrtf(o.x * o.x + o.y * o.y);                     // cessb::clipper_cc(0.95), clipper_cc_impl.cc:74-88
            const float ph = fast_atan2f_lut(o.y, o.x, T);
            const float c = mag < P.clip ? mag : P.clip;
            const float2 sc = sincos_rad(ph);
            P.outc.p[(size_t)b * (P.outc.mask + 1u) + ((uint32_t)st.g & P.outc.mask)] = make_float2(sc.x * c, sc.y * c);
            ++st.g;
            continue;
        }
        float d;
        if (KIND == 1) {
            const float m = sqrtf(s.x * s.x + s.y * s.y);                       // complex_to_mag
            const float o = m * st.gain;                                        // agc2_ff(0.1, 0.1, 1, 1)
            const float tmp = -1.0f + fabsf(o);
            float rate = P.decay;
            if (fabsf(tmp) > st.gain) rate = P.attack;
            st.gain -= tmp * rate;
            if (st.gain < 0.0f) st.gain = 10e-5f;
            if (st.gain > 65536.0f) st.gain = 65536.0f;
            double acc = P.ff0 * (double)o;                                     // iir_filter_ffd({1, -1}, {0, 0.9999})
            acc += P.ff1 * (double)st.iir_x;
            acc += P.fb1 * st.iir_y;
            st.iir_y = acc; st.iir_x = o;
            d = (float)acc * 0.99f;                                             // _audio_gain
        } else {
            const float re = s.x * st.prev.x + s.y * st.prev.y;                 // quadrature_demod_cf
            const float im = s.y * st.prev.x - s.x * st.prev.y;
            d = P.gain * fast_atan2f_lut(im, re, T);
            st.prev = s;
            if (KIND == 2) {
                const float o = d * 0.9f;                                       // _amplify
                double acc = P.ff0 * (double)o;                                 // _de_emph_filter at 200 ksps
                acc += P.ff1 * (double)st.iir_x;
                acc += P.fb1 * st.iir_y;
                st.iir_y = acc; st.iir_x = o;
                d = (float)acc;
            }
        }
        out[(uint32_t)st.g & P.out.mask] = d;
        ++st.g;
    }
    }
    P.st[b] = st;
}
void launch_an_gate(const AnGateParams& p, int kind, int batch, hipStream_t s)
{
    const dim3 g((batch + 63) / 64), t(64);
    if (kind == 0) hipLaunchKernelGGL(k_an_gate<0>, g, t, 0, s, p, batch);
    else if (kind == 1) hipLaunchKernelGGL(k_an_gate<1>, g, t, 0, s, p, batch);
    else if (kind == 2) hipLaunchKernelGGL(k_an_gate<2>, g, t, 0, s, p, batch);
    else hipLaunchKernelGGL(k_an_gate<3>, g, t, 0, s, p, batch);
}

// rational_resampler_fff(I, D) over the gated ring: output q = sum_j taps[ph + j I] x[c - j], ph = q D mod I, c = q D / I
// (one fmaf chain, j ascending).  Outputs of this call: decim_count(g_prev) .. decim_count(g) of the stream.
__global__ __launch_bounds__(256) void k_an_resamp(const AnResampParams P)
{
    extern __shared__ float an_rt[];
    for (int k = threadIdx.x; k < P.nt; k += 256) an_rt[k] = P.taps[k];
    __syncthreads();
    const int b = blockIdx.y;
    uint64_t q0 = P.q0, q1 = P.q0 + P.count;
    if (P.st) { const AnState& st = P.st[b]; q0 = an_decim_count(st.g_prev, P.I, P.D); q1 = an_decim_count(st.g, P.I, P.D); }
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t == 0 && P.port && P.counts) P.counts[b * 4 + 1] = (uint32_t)(q1 - q0);
    const uint64_t q = q0 + t;
    if (q >= q1) return;
    const uint64_t u = q * (uint64_t)P.D;
    const int ph = (int)(u % (uint64_t)P.I);
    const int64_t c = (int64_t)(u / (uint64_t)P.I);
    float a = 0.f;
    for (int j = 0; ph + j * P.I < P.nt; ++j) {
        if (c - j < 0) break;
        a = fmaf(an_rt[ph + j * P.I], an_ringf_at(P.in, b, c - j), a);
    }
    P.out.p[(size_t)b * (P.out.mask + 1u) + ((uint32_t)q & P.out.mask)] = a;
    if (P.port && t < P.port_cap) P.port[(size_t)b * P.port_cap + t] = a;
}
void launch_an_resamp(const AnResampParams& p, uint32_t max_out, int batch, hipStream_t s)
{
    hipLaunchKernelGGL(k_an_resamp, dim3((max_out + 255) / 256, batch), dim3(256), (size_t)p.nt * sizeof(float), s, p);
}

// fft_filter_fff (audio filter at 8 ksps) over the resampler's outputs of this call
__global__ __launch_bounds__(256) void k_an_fir(const AnFirParams P)
{
    extern __shared__ float an_ft[];
    for (int k = threadIdx.x; k < P.nt; k += 256) an_ft[k] = P.taps[k];
    __syncthreads();
    const int b = blockIdx.y;
    const AnState& st = P.st[b];
    uint64_t q0 = an_decim_count(st.g_prev, P.I, P.D), q1 = an_decim_count(st.g, P.I, P.D);
    if (P.cs) { q0 = P.cs[b].g2_prev; q1 = P.cs[b].g2; }        // behind the CTCSS gate: what passed it in this call
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t == 0 && P.port && P.counts) P.counts[b * 4 + 1] = (uint32_t)(q1 - q0);
    const uint64_t q = q0 + t;
    if (q >= q1) return;
    float a = 0.f;
    const int kmax = q + 1 < (uint64_t)P.nt ? (int)(q + 1) : P.nt;
    for (int k = 0; k < kmax; ++k) a = fmaf(an_ft[k], an_ringf_at(P.in, b, (int64_t)q - k), a);
    if (P.out.p) P.out.p[(size_t)b * (P.out.mask + 1u) + ((uint32_t)q & P.out.mask)] = a;
    if (P.port && t < P.port_cap) P.port[(size_t)b * P.port_cap + t] = a;
}
void launch_an_fir(const AnFirParams& p, uint32_t max_out, int batch, hipStream_t s)
{
    hipLaunchKernelGGL(k_an_fir, dim3((max_out + 255) / 256, batch), dim3(256), (size_t)p.nt * sizeof(float), s, p);
}

// cessb::stretcher_cc (stretcher_cc_impl.cc:68-106) + complex_to_real + multiply_const_ff(level): item k of the gated stream divided
// by h = (max(emax max |x[k-2 .. k+2]|, 1) - 1) 2 + 1; whole chunks of 1024 only (an_decim_count with I = 0)
__global__ __launch_bounds__(256) void k_an_stretch(const AnStretchParams P)
{
    const int b = blockIdx.y;
    const AnState& st = P.st[b];
    const uint64_t q0 = an_decim_count(st.g_prev, 0, 1), q1 = an_decim_count(st.g, 0, 1);
    const uint64_t q = q0 + blockIdx.x * 256u + threadIdx.x;
    if (q >= q1) return;
    const float emax = (float)(1 / (sqrt(0.5) / 2));
    float e = 0.0f;
#pragma unroll
    for (int j = -2; j <= 2; ++j) {
        const float2 x = an_ringc_at(P.in, b, (int64_t)q + j);
        const float m = sqrtf(x.x * x.x + x.y * x.y);
        e = m > e ? m : e;
    }
    float h = e * emax;
    h = h > 1.0f ? h : 1.0f;
    h = h - 1.0f;
    h = h * 2.0f;
    h = h + 1.0f;
    const float2 x = an_ringc_at(P.in, b, (int64_t)q);
    P.out.p[(size_t)b * (P.out.mask + 1u) + ((uint32_t)q & P.out.mask)] = (x.x / h) * P.level;
}
void launch_an_stretch(const AnStretchParams& p, uint32_t max_out, int batch, hipStream_t s)
{
    hipLaunchKernelGGL(k_an_stretch, dim3((max_out + 255) / 256, batch), dim3(256), 0, s, p);
}

// analog::ctcss_squelch_ff(8000, tone, 0.01, 8000, 160, true) on the audio resampler's outputs of this call (gr_demod_nbfm.cpp:59-60,
// 97-123): lane per stream, item by item -- three Goertzel recursions, the block decision every `len` items, the squelch_base state
// machine with its raised-cosine envelope (double table), gating: an item that passes goes to ring item g2++.  Arithmetic of
// oracle/orc_analog.c orc_ctcss_squelch_ff.
__global__ __launch_bounds__(64) void k_an_ctcss(const CtcssParams P, int batch)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= batch) return;
    const AnState& st = P.st[b];
    const uint64_t q0 = an_decim_count(st.g_prev, P.I, P.D), q1 = an_decim_count(st.g, P.I, P.D);
    CtcssState c = P.cs[b];
    c.g2_prev = c.g2;
    float* out = P.out.p + (size_t)b * (P.out.mask + 1u);
    for (uint64_t q = q0; q < q1; ++q) {
        const float x = an_ringf_at(P.in, b, (int64_t)q);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float y = x + P.wr[k] * c.d1[k];
            y = y - c.d2[k];
            c.d2[k] = c.d1[k]; c.d1[k] = y;
        }
        if (++c.processed == P.len) {
            float o[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float re = (float)((0.5 * (double)P.wr[k] * (double)c.d1[k] - (double)c.d2[k]) / (double)P.len);
                const float im = (P.wi[k] * c.d1[k]) / (float)P.len;
                o[k] = floorf(100000.0f * sqrtf(re * re + im * im)) / 100000.0f;
                c.d1[k] = c.d2[k] = 0.0f;
            }
            c.processed = 0;
            c.mute = ((double)o[1] < P.level) || o[1] < o[0] || o[1] < o[2];
        }
        switch (c.state) {
        case 0: if (!c.mute) c.state = P.ramp ? 1 : 2; break;
        case 2: if (c.mute) c.state = P.ramp ? 3 : 0; break;
        case 1:
            c.env = P.env[++c.ramped];
            if (c.ramped >= P.ramp) { c.state = 2; c.env = 1.0; }
            break;
        case 3:
            c.env = P.env[--c.ramped];
            if (c.ramped == 0) c.state = 0;
            break;
        }
        if (c.state != 0) { out[(uint32_t)c.g2 & P.out.mask] = (float)((double)x * c.env); ++c.g2; }
    }
    P.cs[b] = c;
}
void launch_an_ctcss(const CtcssParams& p, int batch, hipStream_t s)
{
    hipLaunchKernelGGL(k_an_ctcss, dim3((batch + 63) / 64), dim3(64), 0, s, p, batch);
}

// NBFM: iir_filter_ffd(btaps, ataps, false) de-emphasis + multiply_const_ff(2.0) -> port 1
__global__ __launch_bounds__(64) void k_an_deemph(const AnDeemphParams P, int batch)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= batch) return;
    AnState st = P.st[b];
    uint64_t q0 = an_decim_count(st.g_prev, P.I, P.D), q1 = an_decim_count(st.g, P.I, P.D);
    if (P.cs) { q0 = P.cs[b].g2_prev; q1 = P.cs[b].g2; }
    float x1 = st.de_x; double y1 = st.de_y;
    for (uint64_t q = q0; q < q1; ++q) {
        const float x = an_ringf_at(P.in, b, (int64_t)q);
        double acc = P.ff0 * (double)x;
        acc += P.ff1 * (double)x1;
        acc += P.fb1 * y1;
        y1 = acc; x1 = x;
        const uint64_t t = q - q0;
        if (P.port && t < P.port_cap) P.port[(size_t)b * P.port_cap + t] = (float)acc * 2.0f;
    }
    P.st[b].de_x = x1; P.st[b].de_y = y1;
    if (P.port && P.counts) P.counts[b * 4 + 1] = (uint32_t)(q1 - q0);
}
void launch_an_deemph(const AnDeemphParams& p, int batch, hipStream_t s)
{
    hipLaunchKernelGGL(k_an_deemph, dim3((batch + 63) / 64), dim3(64), 0, s, p, batch);
}

// ---- analogue modulators: gr_mod_nbfm (reference src/gr/gr_mod_nbfm.cpp:26-77).  The chain reuses k_fir_fff (audio filter),
// k_an_resamp (25:4), k_tx_fm, k_fir_ccf, k_scale_c and k_tx_interp_c; new here: audio into a ring, and gain + pre-emphasis.
__global__ __launch_bounds__(256) void k_am_load(const AmLoadParams P)
{
    const int b = blockIdx.y;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= P.count) return;
    P.out.p[(size_t)b * (P.out.mask + 1u) + ((uint32_t)(P.n0 + t) & P.out.mask)] = P.in[(size_t)b * P.in_stride + t];
}
void launch_am_load(const AmLoadParams& p, int batch, hipStream_t s)
{
    if (!p.count) return;
    hipLaunchKernelGGL(k_am_load, dim3((p.count + 255) / 256, batch), dim3(256), 0, s, p);
}
// analog::sig_source_f(fs, GR_SIN_WAVE, f, ampl, offset) into an audio ring (the CW key's tone source, gr_mod_base.cpp:144): GNU Radio's fixed-point NCO
// with its 1024-row sine table (oracle orc_sig_source_f): sample k = (float)(sin_fx(k inc) * ampl) + offset, the phase runs over all samples produced
__global__ __launch_bounds__(256) void k_am_tone(const AmToneParams P)
{
    const int b = blockIdx.y;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= P.count) return;
    const uint32_t u = (uint32_t)((P.k0 + t) * (uint64_t)P.inc);
    float v = P.tab[2 * (u >> 22)] * (float)(u >> 1);
    v = v + P.tab[2 * (u >> 22) + 1];
    float x = (float)((double)v * P.ampl);
    x = x + P.offset;
    P.out.p[(size_t)b * (P.out.mask + 1u) + ((uint32_t)(P.n0 + t) & P.out.mask)] = x;
}
void launch_am_tone(const AmToneParams& p, int batch, hipStream_t s)
{
    if (!p.count) return;
    hipLaunchKernelGGL(k_am_tone, dim3((p.count + 255) / 256, batch), dim3(256), 0, s, p);
}
// multiply_const_ff(gain) -> iir_filter_ffd(btaps, ataps, false): acc = b0 x + b1 x[-1] + fb1 y[-1] in double (fb1 = -a1)
__global__ __launch_bounds__(64) void k_am_iir(const AmIirParams P, int batch)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= batch) return;
    AmIirState st = P.st[b];
    const float* in = P.in.p + (size_t)b * (P.in.mask + 1u);
    float* out = P.out.p + (size_t)b * (P.out.mask + 1u);
    for (uint32_t t = 0; t < P.count; ++t) {
        const uint32_t n = (uint32_t)(P.n0 + t);
        float x = in[n & P.in.mask] * P.gain;
        if (P.tone_tab) {                                                         // _add: the CTCSS tone (sig_source_f, fixed-point NCO + 1024-row sine table)
            const uint32_t u = (uint32_t)((P.tone_k0 + t) * (uint64_t)P.tone_inc) + 0x40000000u;
            float v = P.tone_tab[2 * (u >> 22)] * (float)(u >> 1);
            v = v + P.tone_tab[2 * (u >> 22) + 1];
            x = x + (float)((double)v * P.tone_ampl);
        }
        double acc = P.ff0 * (double)x;
        acc += P.ff1 * (double)st.x1;
        acc += P.fb1 * st.y1;
        st.y1 = acc; st.x1 = x;
        out[n & P.out.mask] = (float)acc;
    }
    P.st[b] = st;
}
void launch_am_iir(const AmIirParams& p, int batch, hipStream_t s)
{
    if (!p.count) return;
    hipLaunchKernelGGL(k_am_iir, dim3((batch + 63) / 64), dim3(64), 0, s, p, batch);
}

// gr_mod_am (reference src/gr/gr_mod_am.cpp:40-45,66-70): agc2_ff(attack, decay, 1, 1) with set_max_gain(1) -> rail_ff(-0.98, 0.98) ->
// multiply_const_ff(0.95): a serial gain recursion per stream (one lane each), arithmetic of oracle/orc_analog.c orc_agc2_ff
__global__ __launch_bounds__(64) void k_am_agc_rail(const AmAgcParams P, int batch)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= batch) return;
    float gain = P.gain[b];
    const float* in = P.in.p + (size_t)b * (P.in.mask + 1u);
    float* out = P.out.p + (size_t)b * (P.out.mask + 1u);
    for (uint32_t t = 0; t < P.count; ++t) {
        const uint32_t n = (uint32_t)(P.n0 + t);
        const float o = in[n & P.in.mask] * gain;
        const float tmp = -P.ref + fabsf(o);
        float rate = P.decay;
        if (fabsf(tmp) > gain) rate = P.attack;
        gain -= tmp * rate;
        if (gain < 0.0f) gain = 10e-5f;
        if (P.max_gain > 0.0f && gain > P.max_gain) gain = P.max_gain;
        float v = o;
        if (v < P.lo) v = P.lo; else if (v > P.hi) v = P.hi;
        out[n & P.out.mask] = v * P.scale;
    }
    P.gain[b] = gain;
}
void launch_am_agc_rail(const AmAgcParams& p, int batch, hipStream_t s)
{
    if (!p.count) return;
    hipLaunchKernelGGL(k_am_agc_rail, dim3((batch + 63) / 64), dim3(64), 0, s, p, batch);
}
// add_ff(audio, sig_source_f(8000, GR_COS_WAVE, 0, 0.5)) -> float_to_complex: the carrier source has frequency 0 (:40), a constant
__global__ __launch_bounds__(256) void k_am_carrier(RingF in, RingC out, uint64_t n0, uint32_t count, float carrier)
{
    const int b = blockIdx.y;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= count) return;
    const uint32_t n = (uint32_t)(n0 + t);
    out.p[(size_t)b * (out.mask + 1u) + (n & out.mask)] = make_float2(in.p[(size_t)b * (in.mask + 1u) + (n & in.mask)] + carrier, 0.0f);
}
void launch_am_carrier(RingF in, RingC out, uint64_t n0, uint32_t count, float carrier, int batch, hipStream_t s)
{
    if (!count) return;
    hipLaunchKernelGGL(k_am_carrier, dim3((count + 255) / 256, batch), dim3(256), 0, s, in, out, n0, count, carrier);
}

// gr_mod_ssb (reference src/gr/gr_mod_ssb.cpp:26-82): float_to_complex -> cessb::clipper_cc(0.95), item by item
__global__ __launch_bounds__(256) void k_am_clip(const AmClipParams P)
{
    __shared__ float T[257];
    for (int k = threadIdx.x; k < 257; k += 256) T[k] = P.atan_tab[k];
    __syncthreads();
    const int b = blockIdx.y;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= P.count) return;
    const uint32_t n = (uint32_t)(P.n0 + t);
    const float re = P.in.p[(size_t)b * (P.in.mask + 1u) + (n & P.in.mask)], im = 0.0f;
    const float mag = sqrtf(re * re + im * im);
    const float ph = fast_atan2f_lut(im, re, T);
    const float c = mag < P.clip ? mag : P.clip;
    const float2 sc = sincos_rad(ph);
    P.out.p[(size_t)b * (P.out.mask + 1u) + (n & P.out.mask)] = make_float2(sc.x * c, sc.y * c);
}
void launch_am_clip(const AmClipParams& p, int batch, hipStream_t s)
{
    if (!p.count) return;
    hipLaunchKernelGGL(k_am_clip, dim3((p.count + 255) / 256, batch), dim3(256), 0, s, p);
}
// cessb::stretcher_cc with its complex output (TX side): item q / h, h from the five-point envelope around q
__global__ __launch_bounds__(256) void k_am_stretch(const AmStretchParams P)
{
    const int b = blockIdx.y;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= P.count) return;
    const uint64_t q = P.q0 + t;
    const float emax = (float)(1 / (sqrt(0.5) / 2));
    float e = 0.0f;
#pragma unroll
    for (int j = -2; j <= 2; ++j) {
        const float2 x = an_ringc_at(P.in, b, (int64_t)q + j);
        const float m = sqrtf(x.x * x.x + x.y * x.y);
        e = m > e ? m : e;
    }
    float h = e * emax;
    h = h > 1.0f ? h : 1.0f;
    h = h - 1.0f;
    h = h * 2.0f;
    h = h + 1.0f;
    const float2 x = an_ringc_at(P.in, b, (int64_t)q);
    P.out.p[(size_t)b * (P.out.mask + 1u) + ((uint32_t)q & P.out.mask)] = make_float2(x.x / h, x.y / h);
}
void launch_am_stretch(const AmStretchParams& p, int batch, hipStream_t s)
{
    if (!p.count) return;
    hipLaunchKernelGGL(k_am_stretch, dim3((p.count + 255) / 256, batch), dim3(256), 0, s, p);
}

}  // namespace qrl
