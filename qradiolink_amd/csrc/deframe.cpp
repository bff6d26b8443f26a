// deframe.cpp — host side of the device deframer (gr_deframer_bb, reference src/gr/gr_deframer_bb.cpp:24-185).
#include "../../include/qrl_hip.h"
#include "engine.hpp"
#include <hip/hip_runtime.h>
#include <memory>
#include <new>
#include <string>

using namespace qrl;
extern int qrl_set_error(int code, const std::string& msg);
struct qrl_ctx { int device; };

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return qrl_set_error(QRL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct qrl_deframer {
    qrl_ctx* ctx = nullptr;
    int type = 1, batch = 1;
    hipStream_t stream = nullptr; bool own_stream = false;
    DeframeState* st = nullptr;
    ~qrl_deframer() {
        if (st) (void)hipFree(st);
        if (own_stream && stream) (void)hipStreamDestroy(stream);
    }
};

extern "C" {

int qrl_deframer_create(qrl_ctx* ctx, int deframer_type, int batch, void* hip_stream, qrl_deframer** out)
{
    if (!ctx || !out) return QRL_ERR_ARG;
    if (deframer_type < 1 || deframer_type > 3) return qrl_set_error(QRL_ERR_ARG, "deframer_type must be 1, 2 or 3 (gr_deframer_bb.cpp:36-47)");
    if (batch < 1) return qrl_set_error(QRL_ERR_ARG, "batch must be >= 1");
    std::unique_ptr<qrl_deframer> h(new (std::nothrow) qrl_deframer);
    if (!h) return QRL_ERR_NOMEM;
    h->ctx = ctx; h->type = deframer_type; h->batch = batch;
    HIPCHK(hipSetDevice(ctx->device));
    if (hip_stream) h->stream = static_cast<hipStream_t>(hip_stream);
    else { HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)); h->own_stream = true; }
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->st), (size_t)batch * sizeof(DeframeState)));
    HIPCHK(hipMemset(h->st, 0, (size_t)batch * sizeof(DeframeState)));
    *out = h.release();
    return QRL_OK;
}
void qrl_deframer_destroy(qrl_deframer* h) { if (h) { (void)hipStreamSynchronize(h->stream); delete h; } }
int qrl_deframer_reset(qrl_deframer* h)
{
    if (!h) return QRL_ERR_ARG;
    HIPCHK(hipMemsetAsync(h->st, 0, (size_t)h->batch * sizeof(DeframeState), h->stream));
    return QRL_OK;
}
int qrl_deframer_process(qrl_deframer* h, const uint8_t* bits, size_t stride, size_t n, const uint32_t* counts, size_t count_stride,
                         uint8_t* out, size_t out_cap, uint32_t* out_counts)
{
    if (!h || !bits || !out || !out_counts) return QRL_ERR_ARG;
    if (n > 0xFFFFFFFFull) return qrl_set_error(QRL_ERR_TOO_BIG, "n too large");
    HIPCHK(hipSetDevice(h->ctx->device));
    DeframeParams p{};
    p.bits = bits; p.stride = stride; p.n = (uint32_t)n; p.counts = counts; p.count_stride = count_stride;
    p.type = h->type; p.buf_len = h->type == 1 ? 64u : h->type == 2 ? 32u : 384u;
    p.st = h->st; p.out = out; p.out_cap = out_cap; p.out_counts = out_counts;
    launch_deframe(p, h->batch, h->stream);
    HIPCHK(hipGetLastError());
    return QRL_OK;
}
int qrl_deframer_sync(qrl_deframer* h)
{
    if (!h) return QRL_ERR_ARG;
    HIPCHK(hipStreamSynchronize(h->stream));
    return QRL_OK;
}

// ---- gr_modem::synchronize on the device ------------------------------------------------------------------------------------
}  // extern "C" (reopened below)

struct qrl_framesync {
    qrl_ctx* ctx = nullptr; int batch = 1, cls = 2; uint32_t bit_buf_len = 64, frame_length = 7;
    hipStream_t stream = nullptr; bool own_stream = false;
    FrameSyncState* st = nullptr; uint8_t* bitbuf = nullptr; size_t bitbuf_stride = 0; uint32_t* activity = nullptr;
    ~qrl_framesync() {
        if (st) (void)hipFree(st);
        if (bitbuf) (void)hipFree(bitbuf);
        if (own_stream && stream) (void)hipStreamDestroy(stream);
    }
};
// mode table of gr_modem::toggleRxMode (src/gr_modem.cpp:203-322) and the sync-word classes of gr_modem::findSync (:1183-1282)
static int framesync_geometry(int modem_type, uint32_t& bits, uint32_t& len)
{
    bits = 64; len = 7; int cls = 2;
    switch (modem_type) {
    case QRL_MODEM_BPSK1K: case QRL_MODEM_2FSK1KFM: case QRL_MODEM_2FSK1K: case QRL_MODEM_GMSK1K: case QRL_MODEM_4FSK1KFM: bits = 32; len = 4; cls = 0; break;
    case 1: case QRL_MODEM_4FSK10KFM: case QRL_MODEM_2FSK10KFM: case QRL_MODEM_GMSK10K: bits = 48 * 8; len = 47; break;
    case 2: bits = 3123 * 8; len = 3122; cls = 1; break;
    case QRL_MODEM_QPSK250K: bits = 1517 * 8; len = 1516; cls = 1; break;
    case QRL_MODEM_4FSK100K: bits = 623 * 8; len = 622; cls = 1; break;
    case QRL_MODEM_M17: bits = 46 * 8; len = 46; cls = 3; break;   // gr_modem.cpp:309-313
    default: break;
    }
    return cls;
}

extern "C" {

int qrl_framesync_create(qrl_ctx* ctx, int modem_type, int batch, void* hip_stream, qrl_framesync** out)
{
    if (!ctx || !out) return QRL_ERR_ARG;
    if (batch < 1) return qrl_set_error(QRL_ERR_ARG, "batch must be >= 1");
    std::unique_ptr<qrl_framesync> h(new (std::nothrow) qrl_framesync);
    if (!h) return QRL_ERR_NOMEM;
    h->ctx = ctx; h->batch = batch;
    h->cls = framesync_geometry(modem_type, h->bit_buf_len, h->frame_length);
    HIPCHK(hipSetDevice(ctx->device));
    if (hip_stream) h->stream = static_cast<hipStream_t>(hip_stream);
    else { HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)); h->own_stream = true; }
    h->bitbuf_stride = (h->bit_buf_len + 15u) & ~15u;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->st), (size_t)batch * sizeof(FrameSyncState)));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->bitbuf), (size_t)batch * h->bitbuf_stride));
    HIPCHK(hipMemset(h->st, 0, (size_t)batch * sizeof(FrameSyncState)));
    *out = h.release();
    return QRL_OK;
}
void qrl_framesync_destroy(qrl_framesync* h) { if (h) { (void)hipStreamSynchronize(h->stream); delete h; } }
int qrl_framesync_reset(qrl_framesync* h)
{
    if (!h) return QRL_ERR_ARG;
    HIPCHK(hipMemsetAsync(h->st, 0, (size_t)h->batch * sizeof(FrameSyncState), h->stream));
    return QRL_OK;
}
int qrl_framesync_frame_bytes(const qrl_framesync* h) { return h ? (int)h->frame_length : 0; }
int qrl_framesync_process(qrl_framesync* h, const uint8_t* bits, size_t stride, size_t n, const uint32_t* counts, size_t count_stride,
                          uint8_t* out, size_t out_cap, uint32_t* out_counts)
{
    if (!h || !bits || !out || !out_counts) return QRL_ERR_ARG;
    if (n > 0xFFFFFFFFull) return qrl_set_error(QRL_ERR_TOO_BIG, "n too large");
    HIPCHK(hipSetDevice(h->ctx->device));
    FrameSyncParams p{};
    p.bits = bits; p.stride = stride; p.n = (uint32_t)n; p.counts = counts; p.count_stride = count_stride;
    p.cls = h->cls; p.bit_buf_len = h->bit_buf_len; p.frame_length = h->frame_length;
    p.st = h->st; p.bitbuf = h->bitbuf; p.bitbuf_stride = h->bitbuf_stride;
    p.out = out; p.out_cap = out_cap; p.out_counts = out_counts; p.activity = h->activity;
    launch_framesync(p, h->batch, h->stream);
    HIPCHK(hipGetLastError());
    return QRL_OK;
}
int qrl_framesync_set_activity_output(qrl_framesync* h, uint32_t* activity)
{
    if (!h) return QRL_ERR_ARG;
    h->activity = activity;
    return QRL_OK;
}
int qrl_framesync_sync(qrl_framesync* h)
{
    if (!h) return QRL_ERR_ARG;
    HIPCHK(hipStreamSynchronize(h->stream));
    return QRL_OK;
}

}  // extern "C"
