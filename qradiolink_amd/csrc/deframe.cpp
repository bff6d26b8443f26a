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

}  // extern "C"
