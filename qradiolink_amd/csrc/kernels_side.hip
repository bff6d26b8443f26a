// kernels_side.hip — the two side outputs of gr_demod_base next to the demodulator chain:
//   k_rssi        rssi_block (reference src/gr/rssi_block.cpp:31-44) on port 0 (the filtered IQ, gr_demod_base.cpp:199-200):
//                 complex_to_mag_squared -> moving_average_ff(2000, 1, 2000) -> single_pole_iir_filter_ff(0.04) -> nlog10_ff
//                 -> multiply_const_ff(10) -> add_const_ff(level)
//   k_fft_fill / k_fft_power / k_fft_shift   rx_fft_c (src/gr/rx_fft.cpp:71-100,113-131): window multiply into the FFT buffer,
//                 volk_32fc_s32f_power_spectrum_32f on the transform, the half-swap of get_fft_data.  The transform itself is
//                 hipFFT (side.cpp).
// [GR-MEM] the GNU Radio blocks are restated from their published behaviour:
//   moving_average_ff  every work() call starts from a fresh sum of the length - 1 history items (ascending index) and then slides
//                      (sum += newest; out = sum * scale; sum -= oldest), at most max_iter = 2000 outputs per call.  The scheduler's
//                      call sizes are not reproducible; the contract here is a saturated stream: calls of exactly max_iter outputs,
//                      i.e. a fresh sum at every absolute output index that is a multiple of 2000 (chunk invariant).
//   single_pole_iir_filter_ff  y = alpha x + (1 - alpha) y_prev with double taps and double state, output rounded to float.
//   nlog10_ff          n log10(x) + k as volk_32f_log2_32f times n / log2f(10) (n = 1, k = 0); log2 = det_log2f (devmath.hpp).
#include "devmath.hpp"
#include "engine.hpp"

namespace qrl {

constexpr int RS_TILE = 64, RS_PITCH = RS_TILE + 1;

// one lane per stream, workgroup = one wave = 64 streams; the new samples arrive as coalesced 64 x 64 tiles through LDS
__global__ __launch_bounds__(64) void k_rssi(const RssiBlockParams P)
{
    __shared__ float tile[64 * RS_PITCH];
    __shared__ uint32_t cnt_s[64];
    __shared__ uint64_t n_s[64];
    const int lane = threadIdx.x, b0 = blockIdx.x * 64, b = b0 + lane;
    const bool active = b < P.batch;
    RssiState st{};
    uint32_t cnt = 0;
    if (active) {
        st = P.st[b];
        cnt = P.counts ? P.counts[(size_t)b * P.count_stride] : P.n;
        if (cnt > P.n) cnt = P.n;
        if (cnt > P.out_cap) cnt = (uint32_t)P.out_cap;
    }
    cnt_s[lane] = cnt; n_s[lane] = st.n;
    uint32_t cmax = cnt;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)cmax, off, 64); cmax = o > cmax ? o : cmax; }
    __syncthreads();
    float* ring = P.ring + (size_t)(active ? b : 0) * RSSI_RING;
    uint32_t phase = (uint32_t)(st.n % 2000u);
    float last = 0.f;
    for (uint32_t t0 = 0; t0 < cmax; t0 += RS_TILE) {
        // |x|^2 of the new samples: row r = stream b0 + r, lane = sample; kept in the power ring for the look-back of 1999
        for (int r = 0; r < 64 && b0 + r < P.batch; ++r) {
            const uint32_t idx = t0 + lane;
            if (idx < cnt_s[r]) {
                const float2 x = P.in[(size_t)(b0 + r) * P.in_stride + idx];
                const float a = x.x * x.x, c = x.y * x.y;
                const float p = a + c;
                tile[r * RS_PITCH + lane] = p;
                P.ring[(size_t)(b0 + r) * RSSI_RING + ((uint32_t)(n_s[r] + idx) & (RSSI_RING - 1))] = p;
            }
        }
        __syncthreads();
        if (active) {
            const uint32_t jn = cnt > t0 ? (cnt - t0 < RS_TILE ? cnt - t0 : RS_TILE) : 0;
            for (uint32_t j = 0; j < jn; ++j) {
                const uint64_t nabs = st.n + t0 + j;
                if (phase == 0) {   // moving_average_ff: fresh sum of the 1999 history items of this work() call
                    float sum = 0.f;
                    for (int k = 1999; k >= 1; --k) sum += nabs >= (uint64_t)k ? ring[(uint32_t)(nabs - k) & (RSSI_RING - 1)] : 0.f;
                    st.sum = sum;
                }
                st.sum += tile[lane * RS_PITCH + j];
                const float ma = st.sum * 1.0f;                                   // scale = 1
                st.sum -= nabs >= 1999u ? ring[(uint32_t)(nabs - 1999u) & (RSSI_RING - 1)] : 0.f;
                const double y = 0.04 * (double)ma + (1.0 - 0.04) * st.prev;      // single_pole_iir<float, float, double>
                st.prev = y;
                float v = det_log2f((float)y) * P.n_log2_10;                      // nlog10_ff
                v = v * 10.0f;                                                    // multiply_const_ff(10)
                v = v + P.level;                                                  // add_const_ff(level)
                tile[lane * RS_PITCH + j] = v;
                last = v;
                phase = phase + 1 == 2000u ? 0u : phase + 1;
            }
        }
        __syncthreads();
        if (P.out) {
            for (int r = 0; r < 64 && b0 + r < P.batch; ++r) {
                const uint32_t idx = t0 + lane;
                if (idx < cnt_s[r]) P.out[(size_t)(b0 + r) * P.out_cap + idx] = tile[r * RS_PITCH + lane];
            }
        }
        __syncthreads();
    }
    if (active) {
        st.n += cnt;
        if (cnt) st.last = last;
        P.st[b] = st;
        if (P.last) P.last[b] = st.last;                                          // probe_signal_f: the latest value
        if (P.out_counts) P.out_counts[b] = cnt;
    }
}
void launch_rssi(const RssiBlockParams& p, hipStream_t s)
{
    hipLaunchKernelGGL(k_rssi, dim3((p.batch + 63) / 64), dim3(64), 0, s, p);
}

// ---- rx_fft_c
// in[i] * window[counter + i] into the FFT input buffer of every stream (rx_fft.cpp:95)
__global__ __launch_bounds__(256) void k_fft_fill(const float2* __restrict__ in, size_t in_stride, uint32_t i0, uint32_t count, const float* __restrict__ win,
                                                  uint32_t counter, float2* __restrict__ buf, uint32_t N)
{
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= count) return;
    const int b = blockIdx.y;
    const float2 x = in[(size_t)b * in_stride + i0 + j];
    const float w = win[counter + j];
    buf[(size_t)b * N + counter + j] = make_float2(x.x * w, x.y * w);
}
// volk_32fc_s32f_power_spectrum_32f(out, X, N, N): 10 log10(|X / N|^2) as (10 / log2(10)) * log2(re^2 + im^2)
__global__ __launch_bounds__(256) void k_fft_power(const float2* __restrict__ X, float* __restrict__ out, uint32_t N, float inv_norm)
{
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= N) return;
    const int b = blockIdx.y;
    const float2 v = X[(size_t)b * N + j];
    const float re = v.x * inv_norm, im = v.y * inv_norm;
    const float a = re * re, c = im * im;
    out[(size_t)b * N + j] = 3.01029995663981209120f * det_log2f(a + c);
}
// get_fft_data: the two halves swapped (rx_fft.cpp:126-127)
__global__ __launch_bounds__(256) void k_fft_shift(const float* __restrict__ pts, float* __restrict__ out, size_t out_stride, uint32_t N)
{
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= N) return;
    const int b = blockIdx.y;
    const uint32_t h = N / 2;
    out[(size_t)b * out_stride + j] = pts[(size_t)b * N + (j < h ? j + (N - h) : j - h)];
}
void launch_fft_fill(const float2* in, size_t in_stride, uint32_t i0, uint32_t count, const float* win, uint32_t counter, float2* buf, uint32_t N, int batch, hipStream_t s)
{
    if (count) hipLaunchKernelGGL(k_fft_fill, dim3((count + 255) / 256, batch), dim3(256), 0, s, in, in_stride, i0, count, win, counter, buf, N);
}
void launch_fft_power(const float2* X, float* out, uint32_t N, int batch, hipStream_t s)
{
    hipLaunchKernelGGL(k_fft_power, dim3((N + 255) / 256, batch), dim3(256), 0, s, X, out, N, 1.0f / (float)N);
}
void launch_fft_shift(const float* pts, float* out, size_t out_stride, uint32_t N, int batch, hipStream_t s)
{
    hipLaunchKernelGGL(k_fft_shift, dim3((N + 255) / 256, batch), dim3(256), 0, s, pts, out, out_stride, N);
}

}  // namespace qrl
