// side.cpp — host side of the two side outputs of gr_demod_base (kernels_side.hip): rssi_block and rx_fft_c.
//   qrl_rssi_*  reference src/gr/rssi_block.cpp:25-50 (make_rssi_block(level), set_level) fed from port 0 of the demodulator
//               (src/gr/gr_demod_base.cpp:199-200: rssi_valve -> rssi_block -> probe_signal_f)
//   qrl_fft_*   reference src/gr/rx_fft.cpp:44-213 (make_rx_fft_c(fftsize, wintype), work, set_enabled, get_fft_data, set_fft_size,
//               set_window_type), instance make_rx_fft_c(32768, WIN_BLACKMAN_HARRIS) src/gr/gr_demod_base.cpp:166,185
// The transform is hipFFT (batched C2C forward, one plan per FFT size); window, power spectrum and half swap are kernels.
#include "../../include/qrl_hip.h"
#include "engine.hpp"
#include "firdes.hpp"
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <cmath>
#include <memory>
#include <new>
#include <string>
#include <vector>

using namespace qrl;
extern int qrl_set_error(int code, const std::string& msg);
struct qrl_ctx { int device; };

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return qrl_set_error(QRL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define FFTCHK(expr)                                                                          \
    do {                                                                                      \
        hipfftResult r_ = (expr);                                                             \
        if (r_ != HIPFFT_SUCCESS) return qrl_set_error(QRL_ERR_HIP, std::string(#expr) + ": hipfft error " + std::to_string((int)r_)); \
    } while (0)

struct qrl_rssi {
    qrl_ctx* ctx = nullptr; int batch = 1; float level = 0.f;
    hipStream_t stream = nullptr; bool own_stream = false;
    RssiState* st = nullptr; float* ring = nullptr;
    ~qrl_rssi() {
        if (st) (void)hipFree(st);
        if (ring) (void)hipFree(ring);
        if (own_stream && stream) (void)hipStreamDestroy(stream);
    }
};

struct qrl_fft {
    qrl_ctx* ctx = nullptr; int batch = 1;
    unsigned fftsize = 0; int wintype = -1;
    hipStream_t stream = nullptr; bool own_stream = false;
    hipfftHandle plan = 0; bool have_plan = false;
    float* win = nullptr; float2* buf = nullptr; float2* spec = nullptr; float* points = nullptr;
    unsigned counter = 0; int push = 0; bool data_ready = false, enabled = false;
    void release() {
        if (have_plan) { (void)hipfftDestroy(plan); have_plan = false; }
        for (void* p : {(void*)win, (void*)buf, (void*)spec, (void*)points}) if (p) (void)hipFree(p);
        win = nullptr; buf = nullptr; spec = nullptr; points = nullptr;
    }
    ~qrl_fft() {
        release();
        if (own_stream && stream) (void)hipStreamDestroy(stream);
    }
};

// gr::fft::window::build(type, ntaps, beta) [GR-MEM]: the cosine-sum windows over M = ntaps - 1, Kaiser by I0, Bartlett, flat top
static double bessel_i0(double x)
{
    double sum = 1.0, term = 1.0;
    for (int k = 1; k < 200; ++k) {
        term *= (x / (2.0 * k)) * (x / (2.0 * k));
        sum += term;
        if (term < 1e-21 * sum) break;
    }
    return sum;
}
static std::vector<float> fft_window(int type, unsigned n, double beta)
{
    std::vector<float> w(n, 1.0f);
    const double M = (double)n - 1.0;
    switch (type) {
    case 0: case 1: case 2: case 3: case 5: return window((Window)type, (int)n);
    case 4: {   // kaiser
        const double ib = 1.0 / bessel_i0(beta);
        for (unsigned i = 0; i < n; ++i) {
            const double r = 2.0 * i / M - 1.0;
            w[i] = (float)(bessel_i0(beta * std::sqrt(1.0 - r * r)) * ib);
        }
        return w;
    }
    case 6:     // bartlett
        for (unsigned i = 0; i < n; ++i) w[i] = (float)(1.0 - std::fabs(2.0 * i / M - 1.0));
        return w;
    default: {  // flat top
        const double sc = 4.6402, c0 = 1.0 / sc, c1 = 1.93 / sc, c2 = 1.29 / sc, c3 = 0.388 / sc, c4 = 0.0322 / sc, pi = 3.14159265358979323846;
        for (unsigned i = 0; i < n; ++i) {
            const double a = 2.0 * pi * i / M;
            w[i] = (float)(c0 - c1 * std::cos(a) + c2 * std::cos(2 * a) - c3 * std::cos(3 * a) + c4 * std::cos(4 * a));
        }
        return w;
    }
    }
}

static int fft_configure(qrl_fft* h, unsigned fftsize, int wintype)
{
    if (wintype < 0 || wintype > 7) wintype = 0;   // rx_fft.cpp:200-203: out of range -> WIN_HAMMING
    HIPCHK(hipSetDevice(h->ctx->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (fftsize != h->fftsize) {
        h->release();
        const size_t nb = (size_t)h->batch * fftsize;
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->win), (size_t)fftsize * sizeof(float)));
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->buf), nb * sizeof(float2)));
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->spec), nb * sizeof(float2)));
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->points), nb * sizeof(float)));
        HIPCHK(hipMemset(h->buf, 0, nb * sizeof(float2)));
        int n[1] = {(int)fftsize};
        FFTCHK(hipfftPlanMany(&h->plan, 1, n, nullptr, 1, (int)fftsize, nullptr, 1, (int)fftsize, HIPFFT_C2C, h->batch));
        h->have_plan = true;
        FFTCHK(hipfftSetStream(h->plan, h->stream));
        h->fftsize = fftsize; h->counter = 0; h->data_ready = false;
        h->wintype = -1;
    }
    if (wintype != h->wintype) {
        const std::vector<float> w = fft_window(wintype, fftsize, 6.76);
        HIPCHK(hipMemcpy(h->win, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice));
        h->wintype = wintype;
    }
    return QRL_OK;
}

extern "C" {

int qrl_rssi_create(qrl_ctx* ctx, int batch, float level, void* hip_stream, qrl_rssi** out)
{
    if (!ctx || !out) return QRL_ERR_ARG;
    if (batch < 1) return qrl_set_error(QRL_ERR_ARG, "batch must be >= 1");
    std::unique_ptr<qrl_rssi> h(new (std::nothrow) qrl_rssi);
    if (!h) return QRL_ERR_NOMEM;
    h->ctx = ctx; h->batch = batch; h->level = level;
    HIPCHK(hipSetDevice(ctx->device));
    if (hip_stream) h->stream = static_cast<hipStream_t>(hip_stream);
    else { HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)); h->own_stream = true; }
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->st), (size_t)batch * sizeof(RssiState)));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->ring), (size_t)batch * RSSI_RING * sizeof(float)));
    HIPCHK(hipMemset(h->st, 0, (size_t)batch * sizeof(RssiState)));
    HIPCHK(hipMemset(h->ring, 0, (size_t)batch * RSSI_RING * sizeof(float)));
    *out = h.release();
    return QRL_OK;
}
void qrl_rssi_destroy(qrl_rssi* h) { if (h) { (void)hipStreamSynchronize(h->stream); delete h; } }
int qrl_rssi_reset(qrl_rssi* h)
{
    if (!h) return QRL_ERR_ARG;
    HIPCHK(hipMemsetAsync(h->st, 0, (size_t)h->batch * sizeof(RssiState), h->stream));
    HIPCHK(hipMemsetAsync(h->ring, 0, (size_t)h->batch * RSSI_RING * sizeof(float), h->stream));
    return QRL_OK;
}
int qrl_rssi_set_level(qrl_rssi* h, float level) { if (!h) return QRL_ERR_ARG; h->level = level; return QRL_OK; }
int qrl_rssi_process(qrl_rssi* h, const float* filtered, size_t stride, size_t n, const uint32_t* counts, size_t count_stride,
                     float* out, size_t out_cap, float* last, uint32_t* out_counts)
{
    if (!h || !filtered) return QRL_ERR_ARG;
    if (n > 0xFFFFFFFFull) return qrl_set_error(QRL_ERR_TOO_BIG, "n too large");
    if (out && out_cap < 1) return qrl_set_error(QRL_ERR_ARG, "out_cap must be >= 1 when out is given");
    HIPCHK(hipSetDevice(h->ctx->device));
    RssiBlockParams p{};
    p.in = reinterpret_cast<const float2*>(filtered); p.in_stride = stride; p.n = (uint32_t)n;
    p.counts = counts; p.count_stride = count_stride;
    p.st = h->st; p.ring = h->ring; p.batch = h->batch;
    p.level = h->level;
    p.n_log2_10 = 1.0f / log2f(10.0f);                         // nlog10_ff: n / log2f(10), n = 1
    p.out = out; p.out_cap = out ? out_cap : (size_t)0xFFFFFFFFu; p.last = last; p.out_counts = out_counts;
    launch_rssi(p, h->stream);
    HIPCHK(hipGetLastError());
    return QRL_OK;
}
int qrl_rssi_sync(qrl_rssi* h) { if (!h) return QRL_ERR_ARG; HIPCHK(hipStreamSynchronize(h->stream)); return QRL_OK; }
void* qrl_rssi_stream(qrl_rssi* h) { return h ? h->stream : nullptr; }

int qrl_fft_create(qrl_ctx* ctx, int batch, unsigned fftsize, int wintype, void* hip_stream, qrl_fft** out)
{
    if (!ctx || !out) return QRL_ERR_ARG;
    if (batch < 1 || fftsize < 2 || fftsize > (1u << 24)) return qrl_set_error(QRL_ERR_ARG, "batch >= 1, 2 <= fftsize <= 2^24");
    std::unique_ptr<qrl_fft> h(new (std::nothrow) qrl_fft);
    if (!h) return QRL_ERR_NOMEM;
    h->ctx = ctx; h->batch = batch;
    HIPCHK(hipSetDevice(ctx->device));
    if (hip_stream) h->stream = static_cast<hipStream_t>(hip_stream);
    else { HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)); h->own_stream = true; }
    if (int r = fft_configure(h.get(), fftsize, wintype)) return r;
    *out = h.release();
    return QRL_OK;
}
void qrl_fft_destroy(qrl_fft* h) { if (h) { (void)hipStreamSynchronize(h->stream); delete h; } }
int qrl_fft_set_enabled(qrl_fft* h, int enabled) { if (!h) return QRL_ERR_ARG; h->enabled = enabled != 0; return QRL_OK; }
int qrl_fft_set_fft_size(qrl_fft* h, unsigned fftsize)
{
    if (!h) return QRL_ERR_ARG;
    if (fftsize < 2 || fftsize > (1u << 24)) return qrl_set_error(QRL_ERR_ARG, "2 <= fftsize <= 2^24");
    if (fftsize == h->fftsize) return QRL_OK;
    return fft_configure(h, fftsize, h->wintype);
}
unsigned qrl_fft_get_fft_size(const qrl_fft* h) { return h ? h->fftsize : 0; }
int qrl_fft_set_window_type(qrl_fft* h, int wintype) { if (!h) return QRL_ERR_ARG; return fft_configure(h, h->fftsize, wintype); }
int qrl_fft_get_window_type(const qrl_fft* h) { return h ? h->wintype : -1; }

// rx_fft_c::work (rx_fft.cpp:71-100) on n new samples of every stream
int qrl_fft_process(qrl_fft* h, const float* iq, size_t stride, size_t n)
{
    if (!h || !iq) return QRL_ERR_ARG;
    if (n > 0xFFFFFFFFull) return qrl_set_error(QRL_ERR_TOO_BIG, "n too large");
    if (h->push > 0 || !h->enabled) return QRL_OK;             // nobody reads: do not fill the buffer
    HIPCHK(hipSetDevice(h->ctx->device));
    const float2* in = reinterpret_cast<const float2*>(iq);
    const unsigned N = h->fftsize;
    size_t i = 0;
    while (i < n) {
        if (h->counter >= N) {
            h->counter = 0;
            FFTCHK(hipfftExecC2C(h->plan, reinterpret_cast<hipfftComplex*>(h->buf), reinterpret_cast<hipfftComplex*>(h->spec), HIPFFT_FORWARD));
            launch_fft_power(h->spec, h->points, N, h->batch, h->stream);
            h->data_ready = true;
            h->push++;
        }
        const size_t chunk = std::min(n - i, (size_t)(N - h->counter));
        launch_fft_fill(in, stride, (uint32_t)i, (uint32_t)chunk, h->win, h->counter, h->buf, N, h->batch, h->stream);
        h->counter += (unsigned)chunk;
        i += chunk;
    }
    HIPCHK(hipGetLastError());
    return QRL_OK;
}
// rx_fft_c::get_fft_data (rx_fft.cpp:113-131): fft_points[b * out_stride + i] (device), *fft_size = 0 when nothing is ready
int qrl_fft_get_fft_data(qrl_fft* h, float* fft_points, size_t out_stride, unsigned* fft_size)
{
    if (!h || !fft_points || !fft_size) return QRL_ERR_ARG;
    h->push = 0;                                                // want more samples in the FFT
    if (!h->data_ready) { *fft_size = 0; return QRL_OK; }
    if (out_stride < h->fftsize) return qrl_set_error(QRL_ERR_ARG, "out_stride < fft size");
    HIPCHK(hipSetDevice(h->ctx->device));
    launch_fft_shift(h->points, fft_points, out_stride, h->fftsize, h->batch, h->stream);
    HIPCHK(hipGetLastError());
    *fft_size = h->fftsize;
    h->data_ready = false;
    return QRL_OK;
}
int qrl_fft_sync(qrl_fft* h) { if (!h) return QRL_ERR_ARG; HIPCHK(hipStreamSynchronize(h->stream)); return QRL_OK; }
void* qrl_fft_stream(qrl_fft* h) { return h ? h->stream : nullptr; }

}  // extern "C"
