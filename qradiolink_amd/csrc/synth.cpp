// synth.cpp — host side of the multi-carrier MMDVM transmitter (reference src/gr/gr_mod_mmdvm_multi2.cpp:30-128):
// per channel int16 -> FM modulator -> LPF -> x0.8 -> 25/24 resampler, then pfb_synthesizer_ccf(10) -> x(1/N) -> bb gain.
#include "../../include/qrl_hip.h"
#include "engine.hpp"
#include "firdes.hpp"
#include <hip/hip_runtime.h>
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

namespace {
template <class T> struct Buf {
    T* p = nullptr;
    ~Buf() { if (p) (void)hipFree(p); }
    int alloc(size_t n) {
        if (hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return QRL_ERR_NOMEM;
        return hipMemset(p, 0, std::max<size_t>(n, 1) * sizeof(T)) == hipSuccess ? QRL_OK : QRL_ERR_HIP;
    }
    int upload(const std::vector<T>& v) {
        int r = alloc(v.size());
        if (r) return r;
        return v.empty() || hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) == hipSuccess ? QRL_OK : QRL_ERR_HIP;
    }
};
uint32_t pow2ge(size_t v) { uint32_t c = 64; while (c < v) c <<= 1; return c; }
}  // namespace

struct qrl_synth {
    qrl_ctx* ctx = nullptr; qrl_synth_config cfg{};
    hipStream_t stream = nullptr; bool own_stream = false;
    int N = 3, J = 0, filt_nt = 0, rs_Jp = 0; float bb_gain = 1.0f; bool single = false; int rs_I = 25, rs_D = 24;
    Buf<float> filt_taps, rs_taps, syn_taps, rA, phase; Buf<float2> twiddle, rB, rC, rD;
    uint32_t m1 = 0, m25 = 0; uint64_t n1 = 0, n25 = 0;
    int port_chan[16];
    std::vector<ZeroRun> zero_runs; Buf<ZeroRun> zero_dev; size_t zero_dev_cap = 0;   // gr_zero_idle_bursts (qrl_synth_add_zero_runs)
    ~qrl_synth() { if (own_stream && stream) (void)hipStreamDestroy(stream); }
    int reset_state() {
        const size_t S = (size_t)cfg.batch * N;
        if (hipMemset(rA.p, 0, S * (m1 + 1) * sizeof(float)) != hipSuccess || hipMemset(rB.p, 0, S * (m1 + 1) * sizeof(float2)) != hipSuccess ||
            hipMemset(rC.p, 0, S * (m1 + 1) * sizeof(float2)) != hipSuccess || hipMemset(rD.p, 0, S * (m25 + 1) * sizeof(float2)) != hipSuccess ||
            hipMemset(phase.p, 0, S * sizeof(float)) != hipSuccess)
            return QRL_ERR_HIP;
        n1 = n25 = 0;
        zero_runs.clear();
        return QRL_OK;
    }
    // the runs that touch [lo, hi) go to the device and are applied to ring r; runs that end before hi are dropped afterwards
    int apply_zero_runs(RingC r, uint64_t lo, uint64_t hi) {
        std::vector<ZeroRun> live;
        for (const ZeroRun& z : zero_runs) if (z.start < hi && z.start + z.count > lo) live.push_back(z);
        if (!live.empty()) {
            if (live.size() > zero_dev_cap) {
                zero_dev_cap = live.size() + 16;
                if (zero_dev.p) { (void)hipFree(zero_dev.p); zero_dev.p = nullptr; }
                int rr = zero_dev.alloc(zero_dev_cap);
                if (rr) return rr;
            }
            if (hipMemcpyAsync(zero_dev.p, live.data(), live.size() * sizeof(ZeroRun), hipMemcpyHostToDevice, stream) != hipSuccess) return QRL_ERR_HIP;
            if (hipStreamSynchronize(stream) != hipSuccess) return QRL_ERR_HIP;   // `live` is a stack vector: the copy must be through
            launch_zero_runs(r, zero_dev.p, (uint32_t)live.size(), lo, hi, stream);
        }
        std::vector<ZeroRun> keep;
        for (const ZeroRun& z : zero_runs) if (z.start + z.count > hi) keep.push_back(z);
        zero_runs.swap(keep);
        return QRL_OK;
    }
};

extern "C" {

int qrl_synth_create(qrl_ctx* ctx, const qrl_synth_config* cfg, qrl_synth** outp)
{
    if (!ctx || !cfg || !outp) return QRL_ERR_ARG;
    if (cfg->num_channels < 1 || cfg->num_channels > 7) return qrl_set_error(QRL_ERR_ARG, "num_channels must be 1..7 (MAX_MMDVM_CHANNELS)");
    if (cfg->batch < 1 || cfg->max_samples < 1) return qrl_set_error(QRL_ERR_ARG, "batch and max_samples must be >= 1");
    std::unique_ptr<qrl_synth> h(new (std::nothrow) qrl_synth);
    if (!h) return QRL_ERR_NOMEM;
    h->ctx = ctx; h->cfg = *cfg; h->N = cfg->num_channels;
    h->single = cfg->single_carrier != 0;
    if (h->single && cfg->num_channels != 1) return qrl_set_error(QRL_ERR_ARG, "single_carrier needs num_channels = 1");
    if (h->single) { h->rs_I = 125; h->rs_D = 12; }   // gr_mod_mmdvm.cpp:43-45
    h->bb_gain = cfg->bb_gain == 0.0f ? 1.0f : cfg->bb_gain;
    const int fw = cfg->filter_width > 0 ? cfg->filter_width : 5000, M = 10;
    HIPCHK(hipSetDevice(ctx->device));
    if (cfg->hip_stream) h->stream = static_cast<hipStream_t>(cfg->hip_stream);
    else { HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)); h->own_stream = true; }
    int r;
    const std::vector<float> ft = low_pass_2(1, 24000, fw, 2000, 60, WIN_BLACKMAN_HARRIS);          // _filter, :52-53
    h->filt_nt = (int)ft.size();
    const std::vector<float> rt = h->single ? low_pass_2(125, 125 * 24000.0, fw, 2000, 60, WIN_BLACKMAN_HARRIS)   // gr_mod_mmdvm.cpp:43-44
                                            : low_pass_2(25, 600000, fw, 2000, 60, WIN_BLACKMAN_HARRIS);          // _resampler 25/24, :50-51
    const int RI = h->rs_I;
    h->rs_Jp = ((int)rt.size() + RI - 1) / RI;
    std::vector<float> rl((size_t)RI * h->rs_Jp, 0.0f);
    for (size_t k = 0; k < rt.size(); ++k) rl[(k % RI) * h->rs_Jp + k / RI] = rt[k];
    const std::vector<float> st = low_pass_2(10, 250000, fw, 2000, 60, WIN_BLACKMAN_HARRIS);        // synthesizer prototype, :88-90
    h->J = ((int)st.size() + M - 1) / M;
    std::vector<float> sl((size_t)h->J * M, 0.0f);
    for (size_t k = 0; k < st.size(); ++k) sl[k] = st[k];
    std::vector<float2> W(M);
    for (int q = 0; q < M; ++q) W[q] = make_float2((float)std::cos(2 * M_PI * q / M), (float)std::sin(2 * M_PI * q / M));
    if ((r = h->filt_taps.upload(ft)) || (r = h->rs_taps.upload(rl)) || (r = h->syn_taps.upload(sl)) || (r = h->twiddle.upload(W))) return r;
    if (synth_lds_bytes(M, h->J) > 160 * 1024) return qrl_set_error(QRL_ERR_ARG, "synthesizer tile does not fit LDS");
    // port map of :103-118: channels 0..3 -> ports 0..3, channels 4, 5, 6 -> ports 9, 8, 7; the other ports are null sources
    for (int p = 0; p < 16; ++p) h->port_chan[p] = -1;
    for (int c = 0, m = 1; c < h->N; ++c) h->port_chan[c <= 3 ? c : 10 - m++] = c;
    const size_t S = (size_t)cfg->batch * h->N;
    const size_t max25 = cfg->max_samples * h->rs_I / h->rs_D + 2;
    h->m1 = pow2ge(cfg->max_samples + h->filt_nt + h->rs_Jp + 64) - 1;
    h->m25 = pow2ge(max25 + h->J + 64) - 1;
    if ((r = h->rA.alloc(S * (h->m1 + 1))) || (r = h->rB.alloc(S * (h->m1 + 1))) || (r = h->rC.alloc(S * (h->m1 + 1))) ||
        (r = h->rD.alloc(S * (h->m25 + 1))) || (r = h->phase.alloc(S)))
        return qrl_set_error(r, "synthesizer buffers");
    *outp = h.release();
    return QRL_OK;
}
void qrl_synth_destroy(qrl_synth* h) { if (h) { (void)hipStreamSynchronize(h->stream); delete h; } }
int qrl_synth_reset(qrl_synth* h)
{
    if (!h) return QRL_ERR_ARG;
    HIPCHK(hipStreamSynchronize(h->stream));
    return h->reset_state();
}
int qrl_synth_add_zero_runs(qrl_synth* h, const qrl_zero_run* runs, size_t n)
{
    if (!h || (!runs && n)) return QRL_ERR_ARG;
    for (size_t i = 0; i < n; ++i) {
        if (runs[i].stream < 0 || runs[i].stream >= h->cfg.batch || runs[i].channel < 0 || runs[i].channel >= h->N)
            return qrl_set_error(QRL_ERR_ARG, "zero run: stream / channel out of range");
        // gr_zero_idle_bursts keeps ONE counter per stream and a tag overwrites it (gr_zero_idle_bursts.cpp:62-69): a run that starts
        // inside another one ends it there -- the zeroed set is [s_i, min(s_i + c_i, s_next)) over the tags in offset order
        ZeroRun z{(uint32_t)(runs[i].stream * h->N + runs[i].channel), 0u, runs[i].start, runs[i].count};
        for (ZeroRun& o : h->zero_runs) {
            if (o.row != z.row) continue;
            if (o.start < z.start && o.start + o.count > z.start) o.count = z.start - o.start;
            else if (z.start < o.start && z.start + z.count > o.start) z.count = o.start - z.start;
        }
        h->zero_runs.push_back(z);
    }
    return QRL_OK;
}
int qrl_synth_set_bb_gain(qrl_synth* h, float g) { if (!h) return QRL_ERR_ARG; h->bb_gain = g; return QRL_OK; }
size_t qrl_synth_out_cap(const qrl_synth* h, size_t n) { return h ? (n * h->rs_I / h->rs_D + 2) * (h->single ? 1 : 10) : 0; }

int qrl_synth_process(qrl_synth* h, const int16_t* in, size_t stride, size_t n, float* iq, size_t out_stride, size_t* produced)
{
    if (!h || (!in && n) || (!iq && n)) return QRL_ERR_ARG;
    if (n > h->cfg.max_samples) return qrl_set_error(QRL_ERR_TOO_BIG, "n exceeds max_samples");
    if (produced) *produced = 0;
    if (n == 0) return QRL_OK;
    HIPCHK(hipSetDevice(h->ctx->device));
    const int B = h->cfg.batch, N = h->N, S = B * N;
    const uint64_t n1_1 = h->n1 + n;
    const uint64_t RI = (uint64_t)h->rs_I, RD = (uint64_t)h->rs_D;
    const uint64_t n25_1 = n1_1 ? ((n1_1 - 1) * RI + RI - 1) / RD + 1 : 0;   // outputs q of the resampler with q*D/I <= n1_1 - 1
    const uint32_t c1 = (uint32_t)n, c25 = (uint32_t)(n25_1 - h->n25);
    S2fInParams sp{}; sp.in = in; sp.in_stride = stride; sp.out = RingF{h->rA.p, h->m1}; sp.q0 = h->n1; sp.count = c1; sp.scale = 32767.0f; sp.level = 1.0f;
    launch_s2f_in(sp, S, h->stream);
    TxFmParams fp{}; fp.in = sp.out; fp.out = RingC{h->rB.p, h->m1}; fp.n0 = h->n1; fp.count = c1;
    fp.k = (float)(2 * M_PI * 12500.0f / 24000.0f); fp.amp = 1.0f; fp.phase = h->phase.p;            // _fm_modulator, :64-66
    launch_tx_fm(fp, S, h->stream);
    if (h->single && !h->zero_runs.empty()) {   // gr_mod_mmdvm.cpp:57-58: zero_idle_bursts between the FM modulator and the filter (24 ksps)
        const int zr = h->apply_zero_runs(fp.out, h->n1, n1_1);
        if (zr) return zr;
    }
    FirCcfParams ff{}; ff.in = fp.out; ff.out = RingC{h->rC.p, h->m1}; ff.q0 = h->n1; ff.count = c1; ff.taps = h->filt_taps.p; ff.nt = h->filt_nt;
    launch_fir_ccf(ff, S, h->stream);
    launch_scale_c(ff.out, h->n1, c1, 0.8f, S, h->stream);                                           // _amplify, :77-79
    if (h->single) launch_scale_c(ff.out, h->n1, c1, h->bb_gain, S, h->stream);                      // gr_mod_mmdvm.cpp:59-61: bb gain BEFORE the resampler
    ResampParams rp{}; rp.in = nullptr; rp.in_ring = ff.out; rp.n0 = h->n1; rp.n = c1;
    rp.out = RingC{h->rD.p, h->m25}; rp.q0 = h->n25; rp.q_count = c25; rp.taps = h->rs_taps.p; rp.I = h->rs_I; rp.D = h->rs_D; rp.Jp = h->rs_Jp;
    if (h->single) { rp.port = reinterpret_cast<float2*>(iq); rp.port_cap = out_stride; }           // the resampler output IS the 250 ksps signal
    launch_resamp(rp, S, h->stream);
    if (h->single) {
        HIPCHK(hipGetLastError());
    if (qrl::take_launch_error()) return QRL_ERR_HIP;
        h->n1 = n1_1; h->n25 = n25_1;
        if (produced) *produced = (size_t)c25;
        return QRL_OK;
    }
    if (!h->zero_runs.empty()) {   // gr_mod_mmdvm_multi2.cpp:108: zero_idle_bursts behind the 25/24 resampler (25 ksps)
        const int zr = h->apply_zero_runs(rp.out, h->n25, n25_1);
        if (zr) return zr;
    }
    SynthParams yp{}; yp.in = rp.out; yp.nch = N; for (int p = 0; p < 16; ++p) yp.port_chan[p] = h->port_chan[p];
    yp.blk0 = h->n25; yp.nblk = c25; yp.taps = h->syn_taps.p; yp.twiddle = h->twiddle.p; yp.M = 10; yp.J = h->J;
    yp.level = 1.0f / (float)N; yp.bb_gain = h->bb_gain;                                             // _divide_level, _bb_gain :91-95
    yp.out = reinterpret_cast<float2*>(iq); yp.out_stride = out_stride; yp.out_cap = out_stride;
    launch_pfb_synth(yp, B, h->stream);
    HIPCHK(hipGetLastError());
    if (qrl::take_launch_error()) return QRL_ERR_HIP;
    h->n1 = n1_1; h->n25 = n25_1;
    if (produced) *produced = (size_t)c25 * 10;
    return QRL_OK;
}
int qrl_synth_sync(qrl_synth* h)
{
    if (!h) return QRL_ERR_ARG;
    HIPCHK(hipStreamSynchronize(h->stream));
    return QRL_OK;
}

}  // extern "C"
