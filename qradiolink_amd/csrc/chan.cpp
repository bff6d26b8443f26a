// chan.cpp — host side of the multi-carrier MMDVM receiver (reference src/gr/gr_demod_mmdvm_multi2.cpp:58-135):
// PFB channelizer -> per channel {rational_resampler_ccf(24,25), fft_filter_ccf, quadrature_demod_cf, level, float_to_short}.
// A handle owns `batch` wideband inputs and produces the channels [channel_first, channel_first + channel_count):
// a multi-GPU job gives every rank the same wideband samples (or its own inputs) and a different channel range.
#include "../../include/qrl_hip.h"
#include "engine.hpp"
#include "firdes.hpp"
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdlib>
#include <cstring>
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

struct qrl_chan {
    qrl_ctx* ctx = nullptr;
    qrl_chan_config cfg{};
    hipStream_t stream = nullptr; bool own_stream = false;
    int M = 10, J = 0, nt = 0, rs_Jp = 0, filt_nt = 0;
    Buf<float> taps, rs_taps, filt_taps, atan_tab; Buf<float2> twiddle;
    Buf<float> ct_a, ct_b, ct_e;   // step-major tap tables of the fused per-channel kernel
    Buf<float2> hist_a, hist_b; uint32_t hist_len = 0; bool flip = false;
    Buf<float2> r1, r2, r3; Buf<float> r4; uint32_t m1 = 0, m2 = 0;
    uint64_t n_in = 0, n1 = 0, n2 = 0;
    float gain = 0, level = 1.0f, rssi_cal = 0.0f;
    bool tail_only = false;   // form 3: only the per-channel chain; its input = 25 ksps channel streams (qrl_chan_process_channels)
    bool xlat2 = false;   // form 2: N freq-xlating FIR decimators 1:N with the PFB prototype in front of the multi2 per-channel chain (BASELINE configs[3])
    hipEvent_t ev_user = nullptr, ev_user2 = nullptr, ev_ext = nullptr;
    // the serial symbol-sync tail (64 waves for 4096 channel streams: latency bound) runs on its own stream so that it overlaps the
    // channelizer and the fused per-channel kernel of the NEXT call; ring r6 holds two calls, ev_tail[slot] guards its reuse
    hipStream_t tail = nullptr; hipEvent_t ev_ff = nullptr, ev_tail[2] = {nullptr, nullptr}; bool tail_valid[2] = {false, false}; uint64_t call_no = 0;
    uint32_t m6 = 0;
    // round 5: the fused per-channel kernel of call k runs on its own stream (`mid`) BESIDE the channelizers of the calls after it (the PFB form on a
    // handle-owned stream only: with a caller's stream the int16 / RSSI outputs stay ordered on that stream).  The channel ring r1 holds ring_calls = 3
    // calls + the tail's look-back: ev_pfb orders the per-channel kernel behind its channelizer, ev_mid[slot] the channelizer of call k + 3 behind the
    // per-channel kernel of call k (the last reader of the ring items it overwrites).  With TWO calls the channelizer of call k + 2 and the per-channel
    // kernel of call k + 1 became ready at the same instant and the step was bimodal (docs/KERNELS.md 10).
    hipStream_t mid = nullptr; hipEvent_t ev_pfb = nullptr, ev_mid[3] = {nullptr, nullptr, nullptr}, ev_user3 = nullptr; bool mid_valid[3] = {false, false, false};
    int ring_calls = 3;   // calls the channel ring holds: the channelizer may run this many calls minus one ahead of the per-channel kernel
    bool opt_serial_tail = false;
    bool mid_used = false, tail_used = false;   // anything ever enqueued there: qrl_chan_stream_wait leaves idle internal streams alone (a marker on an idle low-priority queue held the waiter back ~0.5 ms)
    int opt_legacy_pfb = 0, opt_legacy_tail = 0;   // qrl_chan_set_option
    bool profiling = false; std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;   // qrl_chan_profile: the HBM-facing kernel(s) of each call
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_tail, prof_ss;                      // ... the fused per-channel kernel and the symbol synchroniser (qrl_chan_profile_read_kernels)
    bool xlat = false; int xl_D = 10, xl_nt = 0, xl_S = 0; Buf<float> xl_taps; Buf<float2> xl_rot_lo; std::vector<uint64_t> xl_inc;   // form 1
    bool single = false; int rs_I = 24, rs_D = 25;   // single: gr_demod_mmdvm (one carrier at 250 ksps, 12/125 resampler, no channelizer)
    float* rssi_out = nullptr; size_t rssi_cap = 0; uint32_t* rssi_counts = nullptr;
    // optional 4FSK symbol tail behind every channel (gr_demod_dmr.cpp:62-105 on the 24 ksps channel signal)
    Buf<float> r5, r6, symf_taps, mmse; Buf<SymSyncState> ss; Buf<uint8_t> soft_dummy; int symf_nt = 0; float ss_alpha = 0, ss_beta = 0;
    uint8_t* fsk_bits = nullptr; size_t fsk_bits_cap = 0; float* fsk_const = nullptr; size_t fsk_const_cap = 0; uint32_t* fsk_counts = nullptr;
    int init_ss() {
        std::vector<SymSyncState> s((size_t)cfg.batch * cfg.channel_count);
        for (auto& x : s) { std::memset(&x, 0, sizeof x); x.avg = x.inst = 5.0f; }
        return hipMemcpy(ss.p, s.data(), s.size() * sizeof(SymSyncState), hipMemcpyHostToDevice) == hipSuccess ? QRL_OK : QRL_ERR_HIP;
    }
    size_t zeroed = 0;
    ~qrl_chan() { for (auto* v : {&prof_events, &prof_tail, &prof_ss}) for (auto& e : *v) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
                  if (ev_user2) (void)hipEventDestroy(ev_user2);
                  if (ev_ext) (void)hipEventDestroy(ev_ext);
                  if (ev_ff) (void)hipEventDestroy(ev_ff);
                  for (auto e : ev_tail) if (e) (void)hipEventDestroy(e);
                  for (auto e : ev_mid) if (e) (void)hipEventDestroy(e);
                  if (ev_pfb) (void)hipEventDestroy(ev_pfb);
                  if (ev_user3) (void)hipEventDestroy(ev_user3);
                  if (mid) (void)hipStreamDestroy(mid);
                  if (tail) (void)hipStreamDestroy(tail);
                  if (ev_user) (void)hipEventDestroy(ev_user); if (own_stream && stream) (void)hipStreamDestroy(stream); }
    int reset_state() {
        const size_t S = (size_t)cfg.batch * cfg.channel_count;
        if (hipMemset(hist_a.p, 0, (size_t)cfg.batch * hist_len * sizeof(float2)) != hipSuccess) return QRL_ERR_HIP;
        if (hipMemset(hist_b.p, 0, (size_t)cfg.batch * hist_len * sizeof(float2)) != hipSuccess) return QRL_ERR_HIP;
        if (hipMemset(r1.p, 0, S * (m1 + 1) * sizeof(float2)) != hipSuccess) return QRL_ERR_HIP;
        if (hipMemset(r2.p, 0, S * (m2 + 1) * sizeof(float2)) != hipSuccess) return QRL_ERR_HIP;
        if (hipMemset(r3.p, 0, S * (m2 + 1) * sizeof(float2)) != hipSuccess) return QRL_ERR_HIP;
        if (hipMemset(r4.p, 0, S * (m2 + 1) * sizeof(float)) != hipSuccess) return QRL_ERR_HIP;
        n_in = n1 = n2 = 0; flip = false;
        mid_valid[0] = mid_valid[1] = mid_valid[2] = false;
        if (!ss.p) call_no = 0;
        if (ss.p) {
            if (hipMemset(r5.p, 0, S * (m2 + 1) * sizeof(float)) != hipSuccess || hipMemset(r6.p, 0, S * (m6 + 1) * sizeof(float)) != hipSuccess) return QRL_ERR_HIP;
            tail_valid[0] = tail_valid[1] = false; call_no = 0;
            return init_ss();
        }
        return QRL_OK;
    }
};

extern "C" {

int qrl_chan_create(qrl_ctx* ctx, const qrl_chan_config* cfg, qrl_chan** outp)
{
    if (!ctx || !cfg || !outp) return QRL_ERR_ARG;
    std::unique_ptr<qrl_chan> h(new (std::nothrow) qrl_chan);
    if (!h) return QRL_ERR_NOMEM;
    h->ctx = ctx; h->cfg = *cfg;
    qrl_chan_config& c = h->cfg;
    if (c.num_channels < 1 || c.num_channels > 64) return qrl_set_error(QRL_ERR_ARG, "num_channels must be 1..64");
    if (c.form < 0 || c.form > 3) return qrl_set_error(QRL_ERR_ARG, "form must be 0 (PFB), 1 (legacy freq-xlating), 2 (freq-xlating bank 1:N) or 3 (per-channel chain only)");
    h->xlat = c.form == 1;
    h->xlat2 = c.form == 2;
    h->tail_only = c.form == 3;
    if (h->tail_only) { c.num_channels = 1; c.channel_first = 0; c.channel_count = 1; }
    h->single = c.num_channels == 1 && c.form == 0;
    if (c.channel_count <= 0) { c.channel_first = 0; c.channel_count = c.num_channels; }
    if (c.channel_first < 0 || c.channel_first + c.channel_count > c.num_channels) return qrl_set_error(QRL_ERR_ARG, "bad channel range");
    if (c.batch < 1 || c.max_chunk < (size_t)c.num_channels || (size_t)c.batch * c.channel_count > 65535)
        return qrl_set_error(QRL_ERR_ARG, "bad batch / max_chunk");
    const int M = h->M = (h->xlat || h->xlat2) ? 1 : c.num_channels;   // M = samples consumed per channel-rate instant of the PFB form
    HIPCHK(hipSetDevice(ctx->device));
    if (c.hip_stream) h->stream = static_cast<hipStream_t>(c.hip_stream);
    else {
        int r0;
        if (std::getenv("QRL_CU_CHAN_MAIN")) { if ((r0 = qrl::create_role_stream(&h->stream, 0, "CHAN_MAIN"))) return r0; }
        else HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        h->own_stream = true;
    }
    int r;
    // prototype: low_pass_2(1, fs, 5000, 2000, 60, BH), fs = 25 kHz * M (gr_demod_mmdvm_multi2.cpp:58-60; 250 ksps for M = 10)
    // _filter_width of the reference factories (gr_demod_mmdvm_multi2.cpp:47,58-63; gr_demod_mmdvm.cpp:40-52); 0 = their default call site value
    const double fwp = (!h->xlat && c.filter_width > 0) ? (double)c.filter_width : 5000.0;
    const std::vector<float> proto = low_pass_2(1, 25000.0 * (h->xlat2 ? c.num_channels : M), fwp, 2000, 60, WIN_BLACKMAN_HARRIS);
    h->nt = (int)proto.size(); h->J = (h->nt + M - 1) / M;
    if (!h->xlat2 && chan_lds_bytes(M, h->J) > 160 * 1024) return qrl_set_error(QRL_ERR_ARG, "channelizer tile does not fit LDS");
    std::vector<float> t((size_t)h->J * M, 0.0f);
    for (int k = 0; k < h->nt; ++k) t[k] = proto[k];
    if ((r = h->taps.upload(t))) return r;
    // W[q] = e^{+j 2 pi q / M} rounded to float, made EXACTLY conjugate symmetric (contract, oracle/orc_chains.c orc_chan_twiddles):
    // entries above M/2 mirror those below, and sin(pi) is 0, not the 1.2e-16 libm returns for the rounded argument -- so that the
    // four summation chains of bin M - c are those of bin c with two signs flipped, bit for bit (k_pfb_stream64 computes half the bins)
    std::vector<float2> W(M);
    for (int q = 0; q <= M / 2; ++q) W[q] = make_float2((float)std::cos(2 * M_PI * q / M), (2 * q == M) ? 0.0f : (float)std::sin(2 * M_PI * q / M));
    for (int q = M / 2 + 1; q < M; ++q) W[q] = make_float2(W[M - q].x, -W[M - q].y);
    if ((r = h->twiddle.upload(W))) return r;
    // multi2: :60-61, used as 24/25 resampler.  single carrier: gr_demod_mmdvm.cpp:43-45, 12/125 from MMDVM_SAMPLE_RATE = 250 ksps
    if (h->single) { h->rs_I = 12; h->rs_D = 125; }
    const std::vector<float> rt = h->single ? low_pass_2(12, 12 * 250000.0, fwp, 2000, 60, WIN_BLACKMAN_HARRIS)
                                            : low_pass_2(1, 600000, fwp, 2000, 60, WIN_BLACKMAN_HARRIS);
    const int RI = h->rs_I;
    h->rs_Jp = ((int)rt.size() + RI - 1) / RI;
    std::vector<float> rl((size_t)RI * h->rs_Jp, 0.0f);
    for (size_t k = 0; k < rt.size(); ++k) rl[(k % RI) * h->rs_Jp + k / RI] = rt[k];
    if ((r = h->rs_taps.upload(rl))) return r;
    const bool ct_ok = !h->single && !h->xlat;
    if (h->xlat) {
        // legacy receiver gr_demod_mmdvm_multi.cpp:58-123: per channel rotator_cc(2 pi (-separation) ct / fs) ->
        // rational_resampler_ccf(1, D, low_pass(1, fs, fw, 3500, BH)) at fs = 24 kHz * D (240 ksps, D = 10 in the reference)
        h->xl_D = c.decimation > 0 ? c.decimation : 10;
        const int fw = c.filter_width > 0 ? c.filter_width : 8000, sep = c.channel_separation > 0 ? c.channel_separation : 25000;
        const double fs = 24000.0 * h->xl_D;
        const std::vector<float> xt = low_pass(1, fs, fw, 3500, WIN_BLACKMAN_HARRIS);
        h->xl_nt = (int)xt.size();
        if (!decim_uses_mfma(h->xl_nt, h->xl_D)) return qrl_set_error(QRL_ERR_ARG, "freq-xlating form: decimation must be >= 8 and the tile must fit the LDS");
        h->xl_S = decim_mfma_steps(h->xl_nt, h->xl_D);
        std::vector<float> g((size_t)decim_mfma_hpn(h->xl_nt, h->xl_D), 0.0f);
        for (int k = 0; k < h->xl_nt; ++k) g[(size_t)k + (size_t)(4 * h->xl_S - h->xl_nt + 1)] = xt[k];
        if ((r = h->xl_taps.upload(g))) return r;
        std::vector<float2> lo((size_t)c.channel_count * 512);
        h->xl_inc.resize(c.channel_count);
        for (int cl = 0; cl < c.channel_count; ++cl) {
            const int i = c.channel_first + cl;
            // :89-95: ct = i for i <= 3, 3 - i above (7 channels at most in the reference); more channels: i <= N/2 ? i : i - N
            const int ct = c.num_channels <= 7 ? (i > 3 ? 3 - i : i) : (i <= c.num_channels / 2 ? i : i - c.num_channels);
            const float carrier_offset = (float)(-sep);
            h->xl_inc[cl] = phase_inc_to_turn(2 * M_PI * carrier_offset * ct / (float)fs);
            for (int k = 0; k < 512; ++k) { float sn, cs; sincos_turn_host((uint64_t)k * h->xl_inc[cl], sn, cs); lo[(size_t)cl * 512 + k] = make_float2(cs, sn); }
        }
        if ((r = h->xl_rot_lo.upload(lo))) return r;
        h->rs_I = 1; h->rs_D = h->xl_D;
    }
    if (h->xlat2) {
        // BASELINE configs[3] literal (SURVEY 8(d) "C4 freq-xlating"): channel i = rotator_cc(2 pi (-25000) ct / fs) ->
        // rational_resampler_ccf(1, N, prototype), fs = 25 kHz N, ct = i (i <= N/2) | i - N: the PFB form's channel map, the
        // reference's way of writing a frequency-translating FIR (gr_demod_mmdvm_multi.cpp:62-66,89-96,111-112).  Behind it the
        // per-channel chain of the PFB form (24/25 resampler ... int16, 4FSK tail).
        if (c.num_channels < 8) return qrl_set_error(QRL_ERR_ARG, "form 2: num_channels (= the decimation) must be >= 8");
        h->xl_D = c.num_channels;
        h->xl_nt = h->nt;
        if (!decim_uses_mfma(h->xl_nt, h->xl_D)) return qrl_set_error(QRL_ERR_ARG, "form 2: the decimator tile does not fit the LDS");
        h->xl_S = decim_mfma_steps(h->xl_nt, h->xl_D);
        std::vector<float> g((size_t)decim_mfma_hpn(h->xl_nt, h->xl_D), 0.0f);
        for (int k = 0; k < h->xl_nt; ++k) g[(size_t)k + (size_t)(4 * h->xl_S - h->xl_nt + 1)] = proto[k];
        if ((r = h->xl_taps.upload(g))) return r;
        const double fs = 25000.0 * c.num_channels;
        std::vector<float2> lo((size_t)c.channel_count * 512);
        h->xl_inc.resize(c.channel_count);
        for (int cl = 0; cl < c.channel_count; ++cl) {
            const int i = c.channel_first + cl;
            const int ct = i <= c.num_channels / 2 ? i : i - c.num_channels;
            const float carrier_offset = -25000.0f;
            h->xl_inc[cl] = phase_inc_to_turn(2 * M_PI * carrier_offset * ct / (float)fs);
            for (int k = 0; k < 512; ++k) { float sn, cs; sincos_turn_host((uint64_t)k * h->xl_inc[cl], sn, cs); lo[(size_t)cl * 512 + k] = make_float2(cs, sn); }
        }
        if ((r = h->xl_rot_lo.upload(lo))) return r;
    }
    const std::vector<float> ft = h->xlat ? low_pass(1, 24000, c.filter_width > 0 ? c.filter_width : 8000, 3500, WIN_BLACKMAN_HARRIS)   // legacy :70-74
                                          : low_pass_2(1, 24000, fwp, 2000, 60, WIN_BLACKMAN_HARRIS);    // :62-63
    h->filt_nt = (int)ft.size();
    if ((r = h->filt_taps.upload(ft)) || (r = h->atan_tab.upload(atan_table()))) return r;
    if (ct_ok && chan_tail_supported(h->rs_I, h->rs_D, h->rs_Jp, h->filt_nt, 0) &&
        ((r = h->ct_a.upload(chan_tail_tables(0, rl.data()))) || (r = h->ct_b.upload(chan_tail_tables(1, ft.data()))))) return r;
    h->gain = h->single ? (float)(24000.0f / (2 * M_PI * 10000.0f))                               // gr_demod_mmdvm.cpp:41,48
                        : (float)(24000.0f / (2 * M_PI * 12500.0f));                              // gr_demod_mmdvm_multi2.cpp:80
    h->hist_len = (h->xlat || h->xlat2) ? (uint32_t)(h->xl_nt + h->xl_D) : h->single ? (uint32_t)(h->rs_Jp + h->rs_D + 2) : (uint32_t)(h->J * M);
    // form 3: the history rows hold the channel samples the fused per-channel kernel re-reads in front of a call (its halo), one row per channel stream
    if (h->tail_only) h->hist_len = (chan_tail_lookback() + 1u) & ~1u;
    const size_t S = (size_t)c.batch * c.channel_count;
    const size_t max1 = c.max_chunk / (h->xlat2 ? h->xl_D : M) + 2, max2 = max1 * h->rs_I / h->rs_D + 2;
    // (the fused per-channel kernel recomputes the halo of its first tile from the channel ring: chan_tail_lookback() items in front of a call)
    // (PFB form on a handle-owned stream: TWO calls, the channelizer of call k + 1 writes while the per-channel kernel of call k still reads)
    const bool can_overlap = h->own_stream && !h->single && !h->xlat && !h->xlat2 && !h->tail_only && ct_ok;
    if (const char* e = std::getenv("QRL_CHAN_RING_CALLS")) { const int v = std::atoi(e); if (v == 2 || v == 3) h->ring_calls = v; }
    h->m1 = (h->single || h->xlat) ? 63 : pow2ge((can_overlap ? h->ring_calls : 1) * max1 + h->rs_Jp + 64 + chan_tail_lookback()) - 1;   // the single-carrier chain reads the caller's IQ directly
    h->m2 = pow2ge(max2 + h->filt_nt + 64 + 300) - 1;   // + one rssi_tag_block window
    if ((r = h->hist_a.alloc((size_t)c.batch * h->hist_len)) || (r = h->hist_b.alloc((size_t)c.batch * h->hist_len)) ||
        (r = h->r1.alloc(S * (h->m1 + 1))) || (r = h->r2.alloc(S * (h->m2 + 1))) || (r = h->r3.alloc(S * (h->m2 + 1))) ||
        (r = h->r4.alloc(S * (h->m2 + 1))))
        return qrl_set_error(r, "channelizer buffers");
    if (can_overlap) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        { int r0; if ((r0 = qrl::create_role_stream(&h->mid, lo, "CHAN_MID"))) return r0; }   // a priority of its own (never the hardware queue of the main or the symbol-sync stream), and BELOW the channelizer's:
                                                                                  // the persistent channelizer workgroups of call k + 1 are placed as the tail of call k drains
        HIPCHK(hipEventCreateWithFlags(&h->ev_pfb, hipEventDisableTiming));
        for (auto& e : h->ev_mid) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        const char* env = std::getenv("QRL_CHAN_SERIAL_TAIL");
        h->opt_serial_tail = env && env[0] == '1';
    }
    *outp = h.release();
    return QRL_OK;
}
void qrl_chan_destroy(qrl_chan* h) { if (h) { (void)hipStreamSynchronize(h->stream); if (h->mid) (void)hipStreamSynchronize(h->mid); if (h->tail) (void)hipStreamSynchronize(h->tail); delete h; } }
int qrl_chan_reset(qrl_chan* h)
{
    if (!h) return QRL_ERR_ARG;
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->mid) HIPCHK(hipStreamSynchronize(h->mid));
    if (h->tail) HIPCHK(hipStreamSynchronize(h->tail));
    return h->reset_state();
}
int qrl_chan_set_option(qrl_chan* h, int option, int value)
{
    if (!h) return QRL_ERR_ARG;
    if (option == QRL_CHAN_OPT_LEGACY_PFB) h->opt_legacy_pfb = value == 1 ? 1 : 0;
    else if (option == QRL_CHAN_OPT_LEGACY_TAIL) {
        // the fused per-channel kernel does not fill the intermediate rings the separate kernels read their history from: only before the first samples
        if (h->n_in != 0 || h->n2 != 0) return qrl_set_error(QRL_ERR_STATE, "QRL_CHAN_OPT_LEGACY_TAIL: only before the first call (or after qrl_chan_reset)");
        h->opt_legacy_tail = value != 0;
    }
    else if (option == QRL_CHAN_OPT_SERIAL_TAIL) {
        // every stream drained: the switch needs no ordering between the two forms
        HIPCHK(hipStreamSynchronize(h->stream));
        if (h->mid) HIPCHK(hipStreamSynchronize(h->mid));
        if (h->tail) HIPCHK(hipStreamSynchronize(h->tail));
        h->mid_valid[0] = h->mid_valid[1] = h->mid_valid[2] = false;
        h->opt_serial_tail = value != 0;
    }
    else return qrl_set_error(QRL_ERR_ARG, "unknown channelizer option");
    return QRL_OK;
}
int qrl_chan_set_level(qrl_chan* h, float level) { if (!h) return QRL_ERR_ARG; h->level = level; return QRL_OK; }
int qrl_chan_calibrate_rssi(qrl_chan* h, float level) { if (!h) return QRL_ERR_ARG; h->rssi_cal = level; return QRL_OK; }
int qrl_chan_set_rssi_output(qrl_chan* h, float* rssi, size_t cap, uint32_t* counts)
{
    if (!h) return QRL_ERR_ARG;
    h->rssi_out = rssi; h->rssi_cap = cap; h->rssi_counts = counts;
    return QRL_OK;
}
int qrl_chan_set_4fsk_output(qrl_chan* h, uint8_t* bits, size_t bits_cap, float* constellation, size_t constellation_cap, uint32_t* counts)
{
    if (!h) return QRL_ERR_ARG;
    if (bits && !counts) return qrl_set_error(QRL_ERR_ARG, "4fsk output needs a counts array [streams][4]");
    HIPCHK(hipSetDevice(h->ctx->device));
    if (bits && !h->ss.p) {   // first use: rings, tables and loop state of the symbol tail
        const size_t S = (size_t)h->cfg.batch * h->cfg.channel_count;
        int r;
        const std::vector<float> rrc = root_raised_cosine(1, 24000, 4800, 0.2, 25 * 5);      // gr_demod_dmr.cpp:62-66
        h->symf_nt = (int)rrc.size();
        const size_t max2 = (h->cfg.max_chunk / (h->xlat2 ? h->xl_D : h->M) + 2) * h->rs_I / h->rs_D + 2;
        h->m6 = pow2ge(2 * max2 + h->symf_nt + 64 + 300) - 1;   // two calls: the symbol sync of call k runs beside the kernels of call k + 1
        if (chan_tail_supported(h->rs_I, h->rs_D, h->rs_Jp, h->filt_nt, h->symf_nt) && (r = h->ct_e.upload(chan_tail_tables(2, rrc.data())))) return r;
        if ((r = h->symf_taps.upload(rrc)) || (r = h->mmse.upload(mmse_table())) || (r = h->r5.alloc(S * (h->m2 + 1))) ||
            (r = h->r6.alloc(S * (h->m6 + 1))) || (r = h->ss.alloc(S)) || (r = h->soft_dummy.alloc(64)))
            return qrl_set_error(r, "4fsk tail buffers");
        if (!h->tail) {
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
            { int r0; if ((r0 = qrl::create_role_stream(&h->tail, hi, "CHAN_TAIL"))) return r0; }   // a priority of its own: never the hardware queue of the main stream (engine.cpp, stream creation)
            HIPCHK(hipEventCreateWithFlags(&h->ev_ff, hipEventDisableTiming));
            for (auto& e : h->ev_tail) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        clock_loop_gains((float)(2 * M_PI / 100.0f), 1.0f, 0.2869f, h->ss_alpha, h->ss_beta);  // :70-71
        HIPCHK(hipStreamSynchronize(h->stream));
        if (h->mid) HIPCHK(hipStreamSynchronize(h->mid));
        HIPCHK(hipStreamSynchronize(h->tail));
        if ((r = h->init_ss())) return r;
    }
    h->fsk_bits = bits; h->fsk_bits_cap = bits_cap; h->fsk_const = constellation; h->fsk_const_cap = constellation_cap; h->fsk_counts = counts;
    return QRL_OK;
}
size_t qrl_chan_out_cap(const qrl_chan* h, size_t n) { return h ? (n / (h->xlat2 ? h->xl_D : h->M) + 2) * h->rs_I / h->rs_D + 2 : 0; }

static int chan_process_impl(qrl_chan* h, const float* iq, size_t stride, size_t n, int16_t* out, size_t out_cap, uint32_t* counts,
                             float* chan_out, size_t chan_pitch, int chan_groups);
int qrl_chan_process(qrl_chan* h, const float* iq, size_t stride, size_t n, int16_t* out, size_t out_cap, uint32_t* counts)
{
    if (!h || h->tail_only) return QRL_ERR_ARG;
    return chan_process_impl(h, iq, stride, n, out, out_cap, counts, nullptr, 0, 0);
}
int qrl_chan_channelize(qrl_chan* h, const float* iq, size_t stride, size_t n, float* chan_out, size_t pitch, int groups)
{
    if (!h || !chan_out || groups < 1) return QRL_ERR_ARG;
    if (h->single || h->xlat || h->xlat2 || h->tail_only) return qrl_set_error(QRL_ERR_ARG, "qrl_chan_channelize: PFB form (form 0, num_channels > 1) only");
    if (h->cfg.channel_count % groups) return qrl_set_error(QRL_ERR_ARG, "qrl_chan_channelize: groups must divide the channel count");
    if (pitch < n / (size_t)h->M) return qrl_set_error(QRL_ERR_ARG, "qrl_chan_channelize: pitch < n / num_channels");
    return chan_process_impl(h, iq, stride, n, nullptr, 0, nullptr, chan_out, pitch, groups);
}
int qrl_chan_process_channels(qrl_chan* h, const float* chan_in, size_t pitch, size_t n1, int16_t* out, size_t out_cap, uint32_t* counts)
{
    if (!h || (!chan_in && n1)) return QRL_ERR_ARG;
    if (!h->tail_only) return qrl_set_error(QRL_ERR_ARG, "qrl_chan_process_channels: form 3 handles only");
    return chan_process_impl(h, chan_in, pitch, n1, out, out_cap, counts, nullptr, 0, 0);
}
static int chan_process_impl(qrl_chan* h, const float* iq, size_t stride, size_t n, int16_t* out, size_t out_cap, uint32_t* counts,
                             float* chan_out, size_t chan_pitch, int chan_groups)
{
    if (!h || (!iq && n)) return QRL_ERR_ARG;
    if (n > h->cfg.max_chunk) return qrl_set_error(QRL_ERR_TOO_BIG, "n exceeds max_chunk");
    if ((h->xlat || h->xlat2) && (n & 1)) return qrl_set_error(QRL_ERR_ARG, "n must be even");
    if (n % (size_t)h->M) return qrl_set_error(QRL_ERR_ARG, "n must be a multiple of num_channels (stream_to_streams)");
    if (n == 0) return QRL_OK;
    HIPCHK(hipSetDevice(h->ctx->device));
    (void)qrl::take_launch_error();
    const int B = h->cfg.batch, M = h->M, CC = h->cfg.channel_count, S = B * CC;
    const float2* in = reinterpret_cast<const float2*>(iq);
    const float2* hist_old = h->flip ? h->hist_b.p : h->hist_a.p;
    float2* hist_new = h->flip ? h->hist_a.p : h->hist_b.p;
    // PFB: one output instant per M inputs; form 2: rational_resampler_ccf(1, N) -- output m exists once input m N does
    const uint64_t n1_1 = h->xlat2 ? (h->n_in + n - 1) / (uint64_t)h->xl_D + 1 : (h->n_in + n) / M;
    // PFB form and form 2: the whole per-channel feed-forward chain in one kernel (kernels_chan_tail.hip)
    const bool fused = !h->single && !h->xlat && !h->opt_legacy_tail && h->ct_a.p && (!h->fsk_bits || h->ct_e.p) &&
                       chan_tail_supported(h->rs_I, h->rs_D, h->rs_Jp, h->filt_nt, h->fsk_bits ? h->symf_nt : 0);
    // ts = the stream of the per-channel kernels: `mid` when they overlap the next call's channelizer, the handle's stream otherwise
    const bool use_mid = h->mid && !h->opt_serial_tail && fused && !chan_out && !h->xlat2 && !h->tail_only;
    const hipStream_t ts = use_mid ? h->mid : h->stream;
    // form 3: the call's input ARE the channel samples (rows = channel streams).  The fused kernel reads them where they are (round 5: the copy
    // into the channel ring was 0.86 of the 3.9 ms of a one-rank cluster step); only the separate kernels of QRL_CHAN_OPT_LEGACY_TAIL need the ring
    if (h->tail_only && !fused) launch_ring_load(in, stride, RingC{h->r1.p, h->m1}, h->n1, (uint32_t)n, S, h->stream);
    const int slot2 = (int)(h->call_no & 1), slotr = (int)(h->call_no % (uint64_t)h->ring_calls);
    if (use_mid) h->mid_used = true;
    if (use_mid && h->mid_valid[slotr]) HIPCHK(hipStreamWaitEvent(h->stream, h->ev_mid[slotr], 0));   // per-channel kernel of call k - ring_calls done: the ring items this call's channelizer overwrites are free
    // rssi counts: the fused kernel writes every row's count itself; a fill kernel in front of it sat on the critical path of every step
    // (0.5 ms in the one-rank cluster order: it is dispatched behind the persistent channelizer workgroups of the next call)
    const bool tail_runs = fused && !chan_out && n1_1 > 0 && ((n1_1 - 1) * (uint64_t)h->rs_I + ((uint64_t)h->rs_I - 1)) / (uint64_t)h->rs_D + 1 > h->n2;   // the fused kernel has outputs to produce
    if (h->rssi_out && h->rssi_counts && !tail_runs) HIPCHK(hipMemsetAsync(h->rssi_counts, 0, (size_t)S * sizeof(uint32_t), ts));
    if (h->tail && h->tail_valid[slot2]) HIPCHK(hipStreamWaitEvent(ts, h->ev_tail[slot2], 0));   // symbol sync of call k - 2 done: its half of ring r6 is free
    ChanParams p{};
    p.in = in; p.in_stride = stride; p.n0 = h->n_in; p.n = (uint32_t)n; p.hist = hist_old; p.hist_len = h->hist_len;
    p.out = RingC{h->r1.p, h->m1}; p.m0 = h->n1; p.m_count = (uint32_t)(n1_1 - h->n1);
    p.taps = h->taps.p; p.twiddle = h->twiddle.p; p.M = M; p.J = h->J; p.c_first = h->cfg.channel_first; p.c_count = CC;
    p.legacy = h->opt_legacy_pfb;
    // the HBM-facing kernel(s) = whatever reads the caller's wideband IQ: the PFB, or the per-channel decimators of forms 1 / 2
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (h->profiling && !h->single && !h->tail_only) { HIPCHK(hipEventCreate(&ev0)); HIPCHK(hipEventCreate(&ev1)); HIPCHK(hipEventRecord(ev0, h->stream)); }
    if (chan_out) {   // qrl_chan_channelize: linear rows of chan_pitch items, grouped by destination rank
        p.out = RingC{reinterpret_cast<float2*>(chan_out), 0}; p.out_pitch = chan_pitch; p.row_cpd = (uint32_t)(CC / chan_groups);
    }
    if (!h->single && !h->xlat && !h->xlat2 && !h->tail_only) {
        launch_pfb_chan(p, B, h->stream);
        if (ev1) { HIPCHK(hipEventRecord(ev1, h->stream)); h->prof_events.emplace_back(ev0, ev1); }
    }
    HistParams hp{};
    hp.in = in; hp.in_stride = stride; hp.n0 = h->n_in; hp.n = (uint32_t)n;
    hp.hist_old = hist_old; hp.hist_new = hist_new; hp.hist_len = h->hist_len; hp.rot_enable = 0;
    if (!h->tail_only) { launch_hist_save(hp, B, h->stream); h->flip = !h->flip; }
    if (chan_out) {   // channelizer only: the per-channel chain runs on the rank that owns the channel (qrl_chan_process_channels there)
        HIPCHK(hipGetLastError());
        if (qrl::take_launch_error()) return QRL_ERR_HIP;
        h->n_in += n; h->n1 = n1_1;
        return QRL_OK;
    }
    // per channel chain on S = batch * channel_count streams
    const uint64_t RI = (uint64_t)h->rs_I, RD = (uint64_t)h->rs_D;
    const uint64_t n2_1 = n1_1 ? ((n1_1 - 1) * RI + (RI - 1)) / RD + 1 : 0;   // outputs q with q*D/I <= n1_1 - 1
    const uint32_t c2 = (uint32_t)(n2_1 - h->n2);
    if (h->xlat || h->xlat2) {   // one front-end launch per channel: same input, that channel's rotator, rows b * CC + cl of ring r2 (form 2: r1)
        for (int cl = 0; cl < CC; ++cl) {
            DecimParams dp{};
            dp.in = in; dp.in_stride = stride; dp.n0 = h->n_in; dp.n = (uint32_t)n; dp.hist = hist_old; dp.hist_len = h->hist_len; dp.hist_raw = 1;
            dp.out = h->xlat2 ? RingC{h->r1.p, h->m1} : RingC{h->r2.p, h->m2}; dp.out_row_mul_m1 = (uint32_t)CC - 1u; dp.out_row_add = (uint32_t)cl;
            dp.m0 = h->xlat2 ? h->n1 : h->n2; dp.m_count = h->xlat2 ? (uint32_t)(n1_1 - h->n1) : c2; dp.D = h->xl_D; dp.gtab = h->xl_taps.p; dp.taps = h->xl_taps.p; dp.S = h->xl_S; dp.nt = h->xl_nt;
            dp.rot_enable = 1; dp.rot_acc = 0; dp.rot_inc = h->xl_inc[cl]; dp.rot_nbase = 0; dp.rot_lo = h->xl_rot_lo.p + (size_t)cl * 512;
            if (launch_decim_mfma(dp, B, h->stream)) return qrl_set_error(QRL_ERR_HIP, "freq-xlating front end: hipFuncSetAttribute failed");
        }
        if (ev1) { HIPCHK(hipEventRecord(ev1, h->stream)); h->prof_events.emplace_back(ev0, ev1); }
    }
    if (use_mid) { HIPCHK(hipEventRecord(h->ev_pfb, h->stream)); HIPCHK(hipStreamWaitEvent(ts, h->ev_pfb, 0)); }
    ResampParams rp{};
    if (h->xlat) {
    } else if (h->single) { rp.in = in; rp.in_stride = stride; rp.hist = hist_old; rp.hist_len = h->hist_len; rp.n0 = h->n_in; rp.n = (uint32_t)n; }
    else { rp.in = nullptr; rp.in_ring = RingC{h->r1.p, h->m1}; rp.n0 = h->n1; rp.n = (uint32_t)(n1_1 - h->n1); }
    rp.out = RingC{h->r2.p, h->m2}; rp.q0 = h->n2; rp.q_count = c2; rp.taps = h->rs_taps.p; rp.I = h->rs_I; rp.D = h->rs_D; rp.Jp = h->rs_Jp;
    if (!h->xlat && !fused) launch_resamp(rp, S, h->stream);
    auto rssi = [&](float2* ring) {   // rssi_tag_block: after the filter in multi2 (:126-127), after the resampler in gr_demod_mmdvm (:53-54)
        if (!h->rssi_out) return;
        RssiParams r{}; r.in = RingC{ring, h->m2}; r.j0 = h->n2 / 300; r.count = (uint32_t)(n2_1 / 300 - h->n2 / 300);
        r.calibration = h->rssi_cal; r.out = h->rssi_out; r.cap = h->rssi_cap; r.counts = h->rssi_counts;
        launch_rssi_tag(r, S, h->stream);
    };
    if (h->single) rssi(h->r2.p);
    if (fused) {
        ChanTailParams tp{};
        tp.in = RingC{h->r1.p, h->m1}; tp.q0 = h->n2; tp.count = c2;
        if (h->tail_only) { tp.lin = in; tp.lin_pitch = stride; tp.lin_base = h->n1; tp.lin_n = (uint32_t)n; tp.hist = hist_old; tp.hist_len = h->hist_len; }
        tp.tab_a = h->ct_a.p; tp.tab_b = h->ct_b.p; tp.tab_e = h->ct_e.p; tp.atan_tab = h->atan_tab.p;
        tp.gain = h->gain; tp.gain2 = (float)(24000 / (M_PI / 2 * (float)(24000 / 5))); tp.level = h->level; tp.scale = 32767.0f;
        tp.s16 = out; tp.s16_cap = out_cap; tp.s16_counts = counts;
        if (h->fsk_bits) tp.out_sym = RingF{h->r6.p, h->m6};
        if (h->rssi_out) { tp.rssi = h->rssi_out; tp.rssi_cap = h->rssi_cap; tp.rssi_counts = h->rssi_counts; tp.rssi_cal = h->rssi_cal;
                           tp.tag0 = h->n2 / 300; tp.ntags = (uint32_t)(n2_1 / 300 - h->n2 / 300); }
        hipEvent_t et0 = nullptr, et1 = nullptr;
        if (h->profiling) { HIPCHK(hipEventCreate(&et0)); HIPCHK(hipEventCreate(&et1)); HIPCHK(hipEventRecord(et0, ts)); }
        launch_chan_tail(tp, S, ts);
        if (et1) { HIPCHK(hipEventRecord(et1, ts)); h->prof_tail.emplace_back(et0, et1); }
        if (use_mid) { HIPCHK(hipEventRecord(h->ev_mid[slotr], ts)); h->mid_valid[slotr] = true; }
        if (h->tail_only) {   // the last hist_len channel samples of every row for the next call (k_hist, as for the wideband input of the other forms).
                              // BEHIND the per-channel kernel: the cluster records "this buffer has been read" behind it, so the channelizer that waits for that
                              // record cannot grab the chip between two per-channel kernels (in front of the kernel, this small launch and the persistent
                              // channelizer workgroups became ready together and it waited ~ 0.6 ms for a place: profiles/r05_c4_cluster_one_rank.log)
            HistParams th{};
            th.in = in; th.in_stride = stride; th.n0 = h->n1; th.n = (uint32_t)n;
            th.hist_old = hist_old; th.hist_new = hist_new; th.hist_len = h->hist_len; th.rot_enable = 0;
            launch_hist_save(th, S, h->stream); h->flip = !h->flip;
        }


    } else {
        FirCcfParams fp{};
        fp.in = RingC{h->r2.p, h->m2}; fp.out = RingC{h->r3.p, h->m2}; fp.q0 = h->n2; fp.count = c2; fp.taps = h->filt_taps.p; fp.nt = h->filt_nt;
        launch_fir_ccf(fp, S, h->stream);
        if (!h->single) rssi(h->r3.p);
        QuadDemodParams qp{};
        qp.in = RingC{h->r3.p, h->m2}; qp.out = RingF{h->r4.p, h->m2}; qp.q0 = h->n2; qp.count = c2; qp.gain = h->gain; qp.atan_tab = h->atan_tab.p;
        if (h->fsk_bits) {   // the 4FSK tail's discriminator (gr_demod_dmr.cpp:72-76: 24000 / (pi/2 * 4800)) reads the same items: one pass for both
            qp.out2 = RingF{h->r5.p, h->m2};
            qp.gain2 = (float)(24000 / (M_PI / 2 * (float)(24000 / 5)));
        }
        if (out) { qp.s16 = out; qp.s16_cap = out_cap; qp.s16_level = h->level; qp.s16_scale = 32767.0f; qp.s16_counts = counts; }   // _level + float_to_short in the same pass
        launch_quad_demod(qp, S, h->stream);
        if (h->fsk_bits) {   // gr_demod_dmr.cpp:72-76 behind the channel filter: discriminator (24000 / (pi/2 * 4800), fused above) -> RRC
            FirFffParams f6{}; f6.in = RingF{h->r5.p, h->m2}; f6.out = RingF{h->r6.p, h->m6}; f6.q0 = h->n2; f6.count = c2; f6.taps = h->symf_taps.p; f6.nt = h->symf_nt;
            launch_fir_fff(f6, S, h->stream);
        }
    }
    {
        if (h->fsk_bits) {   // gr_demod_dmr.cpp:70-105: symbol_sync_ff -> level -> phase modulator -> slicer -> dibits, on the RRC output ring
            // on the tail stream, behind this call's feed-forward kernels; the ring slot it reads is rewritten two calls later
            h->tail_used = true;
            HIPCHK(hipEventRecord(h->ev_ff, ts));
            HIPCHK(hipStreamWaitEvent(h->tail, h->ev_ff, 0));
            // (counts[s*4 + 1] and [s*4 + 2] are written for every stream by the kernel; [0] and [3] are not touched: no fill kernel per call)
            SymSyncParams s{};
            s.in = RingF{h->r6.p, h->m6}; s.avail = n2_1; s.soft = RingB{h->soft_dummy.p, 63}; s.st = h->ss.p; s.mmse = h->mmse.p;
            s.alpha = h->ss_alpha; s.beta = h->ss_beta; s.maxp = 5.0f + 0.06f; s.minp = 5.0f - 0.06f;
            s.ted = 0; s.soft_mul = 128.0f; s.soft_add = 128.0f; s.slicer = 1; s.tail = 1; s.tail_scale = 0.9f; s.slim = 2;   // gr_demod_dmr.cpp:73 _level_control (tail_scale); slim = 16-sample windows, 96-thread workgroups (k_symsync_ff<16, 96>)
            s.bits = h->fsk_bits; s.bits_cap = h->fsk_bits_cap;
            s.port = reinterpret_cast<float2*>(h->fsk_const); s.port_cap = h->fsk_const ? h->fsk_const_cap : 0; s.counts = h->fsk_counts;
            hipEvent_t es0 = nullptr, es1 = nullptr;
            if (h->profiling) { HIPCHK(hipEventCreate(&es0)); HIPCHK(hipEventCreate(&es1)); HIPCHK(hipEventRecord(es0, h->tail)); }
            launch_symsync_ff(s, S, h->tail);
            if (es1) { HIPCHK(hipEventRecord(es1, h->tail)); h->prof_ss.emplace_back(es0, es1); }
            const int slot = (int)(h->call_no & 1);
            HIPCHK(hipEventRecord(h->ev_tail[slot], h->tail));
            h->tail_valid[slot] = true;
        }
    }
    ++h->call_no;
    HIPCHK(hipGetLastError());
    if (qrl::take_launch_error()) return QRL_ERR_HIP;
    h->n_in += n; h->n1 = n1_1; h->n2 = n2_1;
    return QRL_OK;
}
int qrl_chan_stream_wait(qrl_chan* h, void* hip_stream)
{
    if (!h) return QRL_ERR_ARG;
    if (!h->ev_user) HIPCHK(hipEventCreateWithFlags(&h->ev_user, hipEventDisableTiming));
    HIPCHK(hipEventRecord(h->ev_user, h->stream));
    HIPCHK(hipStreamWaitEvent(static_cast<hipStream_t>(hip_stream), h->ev_user, 0));
    if (h->tail && h->tail_used) {
        if (!h->ev_user2) HIPCHK(hipEventCreateWithFlags(&h->ev_user2, hipEventDisableTiming));
        HIPCHK(hipEventRecord(h->ev_user2, h->tail));
        HIPCHK(hipStreamWaitEvent(static_cast<hipStream_t>(hip_stream), h->ev_user2, 0));
    }
    if (h->mid && h->mid_used) {
        if (!h->ev_user3) HIPCHK(hipEventCreateWithFlags(&h->ev_user3, hipEventDisableTiming));
        HIPCHK(hipEventRecord(h->ev_user3, h->mid));
        HIPCHK(hipStreamWaitEvent(static_cast<hipStream_t>(hip_stream), h->ev_user3, 0));
    }
    return QRL_OK;
}
int qrl_chan_wait_for(qrl_chan* h, void* hip_stream)
{
    if (!h) return QRL_ERR_ARG;
    if (!h->ev_ext) HIPCHK(hipEventCreateWithFlags(&h->ev_ext, hipEventDisableTiming));
    HIPCHK(hipEventRecord(h->ev_ext, static_cast<hipStream_t>(hip_stream)));
    HIPCHK(hipStreamWaitEvent(h->stream, h->ev_ext, 0));
    if (h->mid && h->mid_used) HIPCHK(hipStreamWaitEvent(h->mid, h->ev_ext, 0));   // (the int16 / RSSI outputs are written on this stream)
    return QRL_OK;
}
void* qrl_chan_stream(qrl_chan* h) { return h ? h->stream : nullptr; }
int qrl_chan_internal_streams(qrl_chan* h, void* out[3])
{
    if (!h || !out) return QRL_ERR_ARG;
    out[0] = h->stream; out[1] = h->tail; out[2] = h->mid;
    return QRL_OK;
}
int qrl_chan_profile(qrl_chan* h, int enable) { if (!h) return QRL_ERR_ARG; h->profiling = enable != 0; return QRL_OK; }
int qrl_chan_profile_read(qrl_chan* h, double* kernel_ms, uint64_t* launches, const char** kernel_name)
{
    if (!h) return QRL_ERR_ARG;
    HIPCHK(hipStreamSynchronize(h->stream));
    double total = 0;
    for (auto& e : h->prof_events) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, e.first, e.second));
        total += ms;
        (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second);
    }
    if (kernel_ms) *kernel_ms = total;
    if (launches) *launches = h->prof_events.size();
    if (kernel_name) *kernel_name = (h->xlat || h->xlat2) ? "k_decim_mfma (one launch per channel, summed)" : h->single ? "k_resamp" : (h->opt_legacy_pfb == 0 && h->M == 64) ? "k_pfb_stream64" : "k_pfb_chan";
    h->prof_events.clear();
    for (auto* v : {&h->prof_tail, &h->prof_ss}) { for (auto& e : *v) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); } v->clear(); }
    return QRL_OK;
}
int qrl_chan_profile_read_kernels(qrl_chan* h, double ms[3], uint64_t launches[3])
{
    if (!h || !ms || !launches) return QRL_ERR_ARG;
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->mid) HIPCHK(hipStreamSynchronize(h->mid));
    if (h->tail) HIPCHK(hipStreamSynchronize(h->tail));
    std::vector<std::pair<hipEvent_t, hipEvent_t>>* sets[3] = {&h->prof_events, &h->prof_tail, &h->prof_ss};
    for (int k = 0; k < 3; ++k) {
        double total = 0;
        for (auto& e : *sets[k]) {
            float t = 0;
            HIPCHK(hipEventElapsedTime(&t, e.first, e.second));
            total += t;
            (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second);
        }
        ms[k] = total; launches[k] = sets[k]->size();
        sets[k]->clear();
    }
    return QRL_OK;
}
int qrl_chan_sync(qrl_chan* h)
{
    if (!h) return QRL_ERR_ARG;
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->mid) HIPCHK(hipStreamSynchronize(h->mid));
    if (h->tail) HIPCHK(hipStreamSynchronize(h->tail));
    return QRL_OK;
}

}  // extern "C"
