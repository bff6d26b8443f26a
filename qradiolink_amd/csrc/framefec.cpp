// framefec.cpp — C ABI of the frame FEC kernels (kernels_framefec.hip): stateless batch calls on a caller-chosen HIP stream.
//   qrl_bptc19696_decode / _encode   CBPTC19696::decode / encode    reference src/MMDVM/BPTC19696.cpp:47-87
//   qrl_m17_decode_frames            M17FrameDecoder::decodeFrame   reference src/M17/M17/M17FrameDecoder.cpp:44-215
#include "../../include/qrl_hip.h"
#include "engine.hpp"
#include <hip/hip_runtime.h>
#include <string>

using namespace qrl;
extern int qrl_set_error(int code, const std::string& msg);
struct qrl_ctx { int device; };

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return qrl_set_error(QRL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

extern "C" {

int qrl_bptc19696_decode(qrl_ctx* ctx, void* hip_stream, const uint8_t* bursts, size_t n, uint8_t* payloads)
{
    if (!ctx || (n && (!bursts || !payloads))) return qrl_set_error(QRL_ERR_ARG, "qrl_bptc19696_decode: null argument");
    HIPCHK(hipSetDevice(ctx->device));
    launch_bptc_decode(bursts, n, payloads, static_cast<hipStream_t>(hip_stream));
    HIPCHK(hipGetLastError());
    return QRL_OK;
}
int qrl_bptc19696_encode(qrl_ctx* ctx, void* hip_stream, const uint8_t* payloads, size_t n, uint8_t* bursts)
{
    if (!ctx || (n && (!bursts || !payloads))) return qrl_set_error(QRL_ERR_ARG, "qrl_bptc19696_encode: null argument");
    HIPCHK(hipSetDevice(ctx->device));
    launch_bptc_encode(payloads, n, bursts, static_cast<hipStream_t>(hip_stream));
    HIPCHK(hipGetLastError());
    return QRL_OK;
}
int qrl_m17_encode_frames(qrl_ctx* ctx, void* hip_stream, const uint8_t* records, size_t n, uint8_t* frames)
{
    if (!ctx || (n && (!frames || !records))) return qrl_set_error(QRL_ERR_ARG, "qrl_m17_encode_frames: null argument");
    HIPCHK(hipSetDevice(ctx->device));
    launch_m17_encode(records, n, frames, static_cast<hipStream_t>(hip_stream));
    HIPCHK(hipGetLastError());
    return QRL_OK;
}
int qrl_m17_decode_frames(qrl_ctx* ctx, void* hip_stream, const uint8_t* frames, size_t n, uint8_t* records)
{
    if (!ctx || (n && (!frames || !records))) return qrl_set_error(QRL_ERR_ARG, "qrl_m17_decode_frames: null argument");
    HIPCHK(hipSetDevice(ctx->device));
    launch_m17_decode(frames, n, records, static_cast<hipStream_t>(hip_stream));
    HIPCHK(hipGetLastError());
    return QRL_OK;
}

}
