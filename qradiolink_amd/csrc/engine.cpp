// engine.cpp — host side of libqrl_hip.so: builds the per-mode kernel pipeline the reference builds
// as a GNU Radio flowgraph (gr_demod_base.cpp:299-828 connects rotator -> resampler -> gr_demod_X),
// owns all device state, and exposes it through the C ABI of include/qrl_hip.h.
// There is NO CPU fallback: without a usable HIP device qrl_init() fails.
#include "../../include/qrl_hip.h"
#include "engine.hpp"
#include "firdes.hpp"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <mutex>
#include <set>
#include <vector>

using namespace qrl;

static thread_local std::string g_last_error;
static int fail(int code, const std::string& msg) { g_last_error = msg; return code; }
int qrl_set_error(int code, const std::string& msg) { return fail(code, msg); }   // shared with tx.cpp

namespace qrl {
static thread_local bool t_launch_error = false;
hipError_t dyn_lds_limit(const void* kernel, int bytes)
{
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) { t_launch_error = true; qrl_set_error(QRL_ERR_HIP, std::string("hipGetDevice: ") + hipGetErrorString(e)); return e; }
    std::lock_guard<std::mutex> g(mu);
    if (done.count({kernel, dev})) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.insert({kernel, dev});
    else { t_launch_error = true; qrl_set_error(QRL_ERR_HIP, std::string("hipFuncSetAttribute(MaxDynamicSharedMemorySize): ") + hipGetErrorString(e)); }
    return e;
}
bool take_launch_error() { const bool r = t_launch_error; t_launch_error = false; return r; }
int create_role_stream(hipStream_t* s, int priority, const char* role)
{
    const char* e = role ? std::getenv((std::string("QRL_CU_") + role).c_str()) : nullptr;
    int first = 0, count = 0;
    hipError_t err;
    if (e && std::sscanf(e, "%d:%d", &first, &count) == 2 && first >= 0 && count > 0 && first + count <= 32) {
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // 256 CUs
        for (int b = 8 * first; b < 8 * (first + count); ++b) mask[b >> 5] |= 1u << (b & 31);
        err = hipExtStreamCreateWithCUMask(s, 8, mask);
    } else {
        err = hipStreamCreateWithPriority(s, hipStreamNonBlocking, priority);
    }
    if (err != hipSuccess) { qrl_set_error(QRL_ERR_HIP, std::string("stream creation: ") + hipGetErrorString(err)); return QRL_ERR_HIP; }
    return QRL_OK;
}
}  // namespace qrl

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(QRL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));          \
    } while (0)

struct qrl_ctx { int device; };

#ifndef QRL_FEC_GATE_US
#define QRL_FEC_GATE_US 30u   // grouped order: head start of the recursion kernel over the decoder (profiles/r04_c5_rx_timeline.log: without it the decoder takes every wave slot first)
#endif
#ifndef QRL_DEV_SKIP
#define QRL_DEV_SKIP 0   // developer builds only (tools/engine_variants.sh): bit 0 no FLL, 1 no fused 2FSK feed-forward kernel, 2 no symbol sync, 3 no decoder launch -- WRONG results, timing experiments on who stretches the front end
#endif
namespace {

template <class T> struct DevBuf {
    T* p = nullptr; size_t n = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t count) {
        if (p) { (void)hipFree(p); p = nullptr; }   // re-designed filters (qrl_demod_set_filter_width) replace their tables
        n = count;
        if (hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) return QRL_ERR_NOMEM;
        if (hipMemset(p, 0, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) return QRL_ERR_HIP;
        return QRL_OK;
    }
    int upload(const std::vector<T>& v) {
        int r = alloc(v.size());
        if (r) return r;
        if (!v.empty() && hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return QRL_ERR_HIP;
        return QRL_OK;
    }
    int zero() { return hipMemset(p, 0, std::max<size_t>(n, 1) * sizeof(T)) == hipSuccess ? QRL_OK : QRL_ERR_HIP; }
};

uint32_t pow2_at_least(size_t v) { uint32_t c = 64; while (c < v) c <<= 1; return c; }
uint64_t decim_count(uint64_t n, int I, int D) { return n ? ((n - 1) * (uint64_t)I + (uint64_t)I - 1) / (uint64_t)D + 1 : 0; }

// polyphase layout for k_decim: taps[p*Jpad + j] = h[p + j*D]
std::vector<float> decim_layout(const std::vector<float>& h, int D, int Jpad)
{
    std::vector<float> t((size_t)D * Jpad, 0.0f);
    for (size_t k = 0; k < h.size(); ++k) t[(k % D) * Jpad + k / D] = h[k];
    return t;
}
std::vector<float> resamp_layout(const std::vector<float>& h, int I, int Jp)
{
    std::vector<float> t((size_t)I * Jp, 0.0f);
    for (size_t k = 0; k < h.size(); ++k) t[(k % I) * Jp + k / I] = h[k];
    return t;
}
std::vector<float2> to_f2(const std::vector<std::complex<float>>& v)
{
    std::vector<float2> r(v.size());
    for (size_t i = 0; i < v.size(); ++i) r[i] = make_float2(v[i].real(), v[i].imag());
    return r;
}

struct DecimStage {
    bool used = false, mfma = false, pl = false, pm = false;
    int D = 1, Jpad = 0, variant = DECIM_R4_J12, nt = 0, S = 0;
    DevBuf<float> taps;
    DevBuf<float2> edge, edge_b; uint32_t edge_len = 0;   // phase-lane kernels: per-stream scratch for the call's edge outputs (two: staged a call ahead)
    int alloc_edge(int B, bool two = false) {
        if (!pl && !pm) return QRL_OK;
        edge_len = (uint32_t)(pm ? decim_pm_edge_len(nt, D) : decim_pl_edge_len(nt, D));
        if (!edge_len) return QRL_OK;
        if (int r = edge.alloc((size_t)B * edge_len)) return r;
        return two ? edge_b.alloc((size_t)B * edge_len) : QRL_OK;
    }
    int plan(const std::vector<float>& h, int D_) {
        used = true; D = D_; nt = (int)h.size();
        if (decim_uses_pm(nt, D)) {   // phase-major matrix-pipe kernel (the 1:50 first stages)
            pm = true;
            return taps.upload(decim_pm_layout(h, D));
        }
        if (decim_uses_pl(nt, D)) {   // register-resident phase-lane kernel (the 100:1 front end)
            pl = true;
            return taps.upload(decim_pl_layout(h, D));
        }
        if (decim_uses_mfma(nt, D)) {
            // zero-padded tap vector the MFMA A operands are read from: hp[k + (4S - nt + 1)] = h[k]
            mfma = true;
            S = decim_mfma_steps(nt, D);
            std::vector<float> g((size_t)decim_mfma_hpn(nt, D), 0.0f);
            for (int k = 0; k < nt; ++k) g[(size_t)k + (size_t)(4 * S - nt + 1)] = h[k];
            return taps.upload(g);
        }
        const int J = (nt + D - 1) / D;
        const size_t kLds2 = 80 * 1024;  // two workgroups per CU
        auto pad = [&](int v) { const int jc = decim_jc(v); return (J + jc - 1) / jc * jc; };
        variant = -1;
        if (J <= 10 && decim_lds_bytes(D, pad(DECIM_R2_J10), DECIM_R2_J10) <= kLds2) variant = DECIM_R2_J10;
        else if (J > 36 && J <= 44 && decim_lds_bytes(D, 44, DECIM_R4_J44) <= kLds2) variant = DECIM_R4_J44;
        else if (decim_lds_bytes(D, pad(DECIM_R4_J12), DECIM_R4_J12) <= kLds2) variant = DECIM_R4_J12;
        else variant = DECIM_R1_J14;
        Jpad = pad(variant);
        if (decim_lds_bytes(D, Jpad, variant) > 160 * 1024) return QRL_ERR_ARG;
        return taps.upload(decim_layout(h, D, Jpad));
    }
    uint32_t lookback() const { return pm ? decim_pm_lookback(nt, D) : pl ? (uint32_t)(((nt + D - 1) / D + 1) * D) : mfma ? (uint32_t)(nt + D) : (uint32_t)(Jpad * D); }
    int launch(DecimParams& p, int B, hipStream_t s, int parity = 0) const {
        p.nt = nt;
        float2* e = parity && edge_b.p ? edge_b.p : edge.p;
        if (pm) { p.pl_taps = taps.p; p.pl_edge = e; p.pl_edge_stride = edge_len; p.pl_edge_cap = edge_len; return launch_decim_pm(p, B, s); }
        if (pl) { p.pl_taps = taps.p; p.pl_edge = e; p.pl_edge_stride = edge_len; p.pl_edge_cap = edge_len; return launch_decim_pl(p, B, s); }
        if (mfma) { p.gtab = taps.p; p.S = S; return launch_decim_mfma(p, B, s); }
        launch_decim(p, B, variant, s);
        return 0;
    }
};

}  // namespace

struct qrl_demod {
    qrl_ctx* ctx = nullptr;
    qrl_demod_config cfg{};
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // the serial tail (symbol sync + Viterbi: a handful of waves) runs on its own stream so that it overlaps the
    // HBM-facing kernels of the NEXT call instead of idling 250 CUs
    hipStream_t tail = nullptr;
    // QPSK / BPSK / 4FSK-discriminator families: the recursive chain (k_qpsk_*: latency bound, 64 streams per workgroup) runs on
    // `tail`, the Viterbi decoder on `fecs`: call k's decoder, call k + 1's recursion and call k + 2's front end run side by side.
    // The rings between them hold two calls; ev_q / ev_fec guard their reuse two calls later.
    hipStream_t fecs = nullptr;
    hipEvent_t ev_q[2] = {nullptr, nullptr}, ev_fec[2] = {nullptr, nullptr}; bool q_valid[2] = {false, false};
    // GROUPED ORDER (gr_demod_qpsk chain whose recursion kernel has a workgroup for at least every second CU): k_qpsk_pipe4 is a serial
    // walk, one workgroup of 6 waves and 137 KB of LDS per 64 streams.  Left to the three streams, the front end of call k + 1 (tens of
    // thousands of small workgroups) and the decoder of call k - 1 (a wave per two trellises, for its whole run) take every LDS byte
    // and wave slot the moment they free up, and the recursion of call k is only placed once both have drained: (front end || decoder)
    // 2.6 ms, then the recursion alone 1.9 ms (profiles/r04_c5_rx_timeline.log).  Grouped: the front end of call k + 1 waits for the
    // recursion of call k, and the decoder of call k - 1 is LAUNCHED with the recursion of call k (behind the same front-end event),
    // which leaves front end alone -> recursion || decoder.  The deferred launch is flushed by everything that waits for results
    // (qrl_demod_sync, qrl_demod_stream_wait, reset, destroy), so a caller never sees the difference.
    bool grouped = false, grouped_capable = false, fec_deferred = false; FecParams fec_pending{}; int fec_pending_slot = 0;
    DevBuf<uint64_t> qp_snap;   // [2][B] symbols produced up to the end of call k (slot k & 1): what that call's decoder may read
    hipEvent_t ev_ff = nullptr, ev_tail = nullptr;
    // HELPER STREAM of the front end (round 6): k_hist (the rotated tail of this call's IQ, kept for the next call) and k_pl_edge_stage (the
    // next call's edge scratch: that history + the head of the next buffer) read the caller's buffers only, yet they sat between two front-end
    // launches on the handle's stream -- 0.2 - 0.29 ms of C1's 8 ms step (profiles/r06_c1_helper_stream.log).  On `pre` they run BESIDE the
    // front end: edge(k) behind hist(k - 1); hist(k) behind the front end of call k - 1 (the last reader of the history buffer it overwrites);
    // the front end of call k waits for ev_pre.  The history and the edge scratch are double buffers.
    // Only with QRL_OPT_INPUT_RESIDENT: the helpers then read a call's IQ WITHOUT waiting for what the caller queued on the handle's stream before the call.
    hipStream_t pre = nullptr; hipEvent_t ev_pre = nullptr, ev_fe[2] = {nullptr, nullptr}; bool fe_valid[2] = {false, false}; bool pre_pending = false;
    bool input_resident = false;
    hipEvent_t ev_user[4] = {nullptr, nullptr, nullptr, nullptr};   // qrl_demod_stream_wait
    bool tail_pending = false;
    // overlapped mode (2FSK / GMSK / 4FSK families): everything behind the first decimated ring runs on the tail stream while
    // the front end of the NEXT call already runs on the main stream; ring s2 holds two calls, ev_tail2 guards its reuse
    bool qpsk_fll = false, fsk4_disc = false;
    bool m17 = false;   // F_DMR family, gr_demod_m17 variant: channel filter behind the resampler (port 0), mod-M&M TED, no level control
    DevBuf<float2> s2g, disc4_taps; DevBuf<float> sym4_taps; int disc4_nt = 0, sym4_nt = 0;   // 4FSK non-FM branch
    bool fll_slim = false;   // QRL_OPT_FLL_SLIM: single-wave FLL workgroups (3 KB of LDS) that fit beside four front-end workgroups
    bool d2f_capable = false, d2f = false; DevBuf<float> d2f_taps;   // 1:2 decimator + shaping filter in one kernel (k_dec2_fir)
    bool overlap = false, overlap_capable = false; hipEvent_t ev_tail2[2] = {nullptr, nullptr}; bool tail2_valid[2] = {false, false}; uint64_t call_no = 0;
    enum Family { F_2FSK, F_GMSK, F_QPSK, F_DMR, F_4FSK, F_BPSK, F_DSSS, F_ANALOG } fam = F_2FSK;
    int branches = 2;

    // derived chain parameters (gr_demod_2fsk.cpp:39-63, gr_demod_gmsk.cpp:39-63)
    int fe_decim = 1, interp = 1, decim = 1, target = 0, sps_eff = 0;
    bool fm = false;

    // stage objects
    DecimStage fe;      // gr_demod_base resampler (device rate >= 2 Msps)
    DecimStage first;   // per-mode _resampler when interp == 1
    // time-domain scope tap (gr_demod_base.cpp:62-63, 1115-1147, 988-1018): _demod_valve -> rational_resampler_ccf(1, 10, low_pass(1, 1e6,
    // 50000, 25000, HAMMING)) -> gr_sample_sink; off until qrl_demod_set_time_domain_output gives it a buffer
    DecimStage scope; DevBuf<float2> s_scope; uint32_t scope_mask = 0; uint64_t n_scope = 0; int scope_D = 10;
    float2* scope_out = nullptr; size_t scope_cap = 0; uint32_t* scope_counts = nullptr;
    DevBuf<float> rs_taps; int rs_Jp = 0;  // per-mode _resampler when interp > 1
    DevBuf<float> filt_taps; int filt_nt = 0;
    DevBuf<float> symf_taps; int symf_nt = 0;
    DevBuf<float2> disc_up, disc_lo; int disc_nt = 0;
    DevBuf<float> ff_tf, ff_ts; DevBuf<float2> ff_up, ff_lo;   // zero-padded copies for the fused 2FSK kernel (k_2fsk_ff)
    DevBuf<float2> fll_lo, fll_up; float fll_alpha = 0, fll_beta = 0, fll_maxf = 0;
    DevBuf<float> atan_tab, mmse_tab;
    float demod_gain = 0;
    float ss_alpha = 0, ss_beta = 0, ss_maxp = 0, ss_minp = 0;

    // rotator (gr_demod_base.cpp:57,1220-1225): exact 2^-64-turn NCO
    uint64_t rot_inc = 0, rot_acc = 0, rot_nbase = 0;
    DevBuf<float2> rot_lo;

    // rings and state
    DevBuf<float2> hist_a, hist_b; uint32_t hist_len = 0; bool hist_flip = false;
    DevBuf<float2> s1, s2, s2l, s2f; DevBuf<float> s2d, s3; DevBuf<uint8_t> soft;
    uint32_t s1_mask = 0, s2_mask = 0, soft_mask = 0;
    DevBuf<FllState> fll_st; DevBuf<SymSyncState> ss_st; DevBuf<FecState> fec_st;
    // a37b: gr_dmr_dmo_sink behind port 3 of gr_demod_dmr (qrl_demod_set_dmo_output)
    DevBuf<DmoState> dmo_st; DevBuf<uint32_t> dmo_golay; uint8_t* dmo_out = nullptr; uint32_t dmo_cap = 0; uint32_t* dmo_counts = nullptr;
    // QPSK (gr_demod_qpsk.cpp:97-126)
    DevBuf<QpskState> qp_st; DevBuf<float> tanh_tab;
    float c1_alpha = 0, c1_beta = 0, c2_alpha = 0, c2_beta = 0; float2 qp_rot{};
    DevBuf<uint32_t> counts_scratch;
    // DSSS mode (gr_demod_dsss.cpp:30-111): behind the 1:50 stage (ring s2, 20 ksps) a 13/50 resampler to 5 200 samples/s, Costas,
    // channel filter, agc2, Barker-13 matched filter (16 symbols/s), clock recovery + Costas (kernels_dsss.hip)
    DevBuf<float> ds_rs, ds_filt, ds_mf; int ds_Jp = 0, ds_nf = 0;
    DevBuf<float2> ds_ra, ds_rb, ds_rc, ds_rd, ds_sym; uint32_t ds_mask = 0, ds_sym_mask = 0;
    DevBuf<DsssState> ds_st; DevBuf<DsssTailState> ds_tail;
    float ds_a1 = 0, ds_b1 = 0, ds_a2 = 0, ds_b2 = 0;
    uint64_t n5 = 0, nsy = 0;   // items so far at 5 200 samples/s, matched-filter outputs so far
    int dsss_stages(uint64_t n2_0, uint64_t n2_1, const qrl_demod_out* out, uint32_t* counts, bool side);
    // analogue voice receivers (gr_demod_nbfm / gr_demod_am / gr_demod_wbfm): kernels_analog.hip
    int an_kind = 0; bool an_lsb = false;                        // 0 NBFM, 1 AM, 2 WBFM, 3 SSB (an_lsb: lower sideband)
    DevBuf<float2> an_c1;                                        // SSB: clipped complex items behind the gate
    DevBuf<float2> an_filt_c; int an_nfc = 0;                    // AM channel filter (complex taps)
    DevBuf<float> an_env, an_rtaps, an_ftaps; int an_ramp = 0, an_nr = 0, an_nf = 0, an_I = 2, an_D = 5;
    DevBuf<float> an_f1, an_f2, an_f3; uint32_t an_m1 = 0, an_m2 = 0;
    DevBuf<AnState> an_st;
    // gr_demod_nbfm::set_ctcss: ctcss_squelch_ff between audio resampler and audio filter, band-pass audio filter while it is on
    float ctcss_tone = 0.0f; DevBuf<CtcssState> an_cs; DevBuf<float> an_f4, an_ftaps_ct; DevBuf<double> an_env_ct; int an_nf_ct = 0;
    float ct_wr[3] = {0, 0, 0}, ct_wi[3] = {0, 0, 0};
    double an_threshold = 1e-14, an_ff[2] = {0, 0}, an_fb1 = 0, an_de_ff[2] = {0, 0}, an_de_fb1 = 0;
    float an_gain = 1.f, an_attack = 0.1f, an_decay = 0.1f;
    int analog_stages(uint64_t n2_0, uint64_t n2_1, const qrl_demod_out* out, uint32_t* counts, bool side);
    float an_if_gain = 0.9f;
    uint64_t n_in = 0, n1 = 0, n2 = 0;  // items so far: device rate, 1 Msps, target rate
    bool profiling = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;

    ~qrl_demod() {
        for (auto& e : prof_events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
        if (ev_ff) (void)hipEventDestroy(ev_ff);
        if (ev_tail) (void)hipEventDestroy(ev_tail);
        for (auto e : ev_user) if (e) (void)hipEventDestroy(e);
        for (auto e : ev_tail2) if (e) (void)hipEventDestroy(e);
        for (auto e : ev_q) if (e) (void)hipEventDestroy(e);
        for (auto e : ev_fec) if (e) (void)hipEventDestroy(e);
        if (ev_pre) (void)hipEventDestroy(ev_pre);
        for (auto e : ev_fe) if (e) (void)hipEventDestroy(e);
        if (pre) (void)hipStreamDestroy(pre);
        if (fecs) (void)hipStreamDestroy(fecs);
        if (tail) (void)hipStreamDestroy(tail);
        if (own_stream && stream) (void)hipStreamDestroy(stream);
    }

    int upload_rot_table() {
        std::vector<float2> lo(512);
        for (int r = 0; r < 512; ++r) { float s, c; sincos_turn_host((uint64_t)r * rot_inc, s, c); lo[r] = make_float2(c, s); }
        if (!rot_lo.p) return rot_lo.upload(lo);
        return hipMemcpy(rot_lo.p, lo.data(), 512 * sizeof(float2), hipMemcpyHostToDevice) == hipSuccess ? QRL_OK : QRL_ERR_HIP;
    }
    int flush_fec(bool behind_front_end) {
        if (!fec_deferred) return QRL_OK;
        fec_deferred = false;
        HIPCHK(hipStreamWaitEvent(fecs, ev_q[fec_pending_slot], 0));
        if (behind_front_end) {   // starts with the recursion of the next call, not beside its front end -- and a moment AFTER it (k_fec_gate)
            HIPCHK(hipStreamWaitEvent(fecs, ev_ff, 0));
            launch_fec_gate(QRL_FEC_GATE_US, fecs);
        }
        launch_fec(fec_pending, cfg.batch, fecs);
        HIPCHK(hipEventRecord(ev_fec[fec_pending_slot], fecs));
        return QRL_OK;
    }
    int sync_all() {
        if (int r = flush_fec(false)) return r;
        HIPCHK(hipStreamSynchronize(stream));
        HIPCHK(hipStreamSynchronize(tail));
        HIPCHK(hipStreamSynchronize(fecs));
        if (pre) HIPCHK(hipStreamSynchronize(pre));
        return QRL_OK;
    }
    bool loops_family() const { return fam == F_QPSK || fam == F_BPSK || fsk4_disc; }
    int init_state();
    int build();
    int process(const float* iq, size_t stride, size_t n, const qrl_demod_out* out);
};

int qrl_demod::init_state()
{
    int r;
    for (auto* b : {&hist_a, &hist_b, &s1, &s2, &s2l, &s2f}) if (b->p && (r = b->zero())) return r;
    for (auto* b : {&s2d, &s3}) if (b->p && (r = b->zero())) return r;
    if (s_scope.p && (r = s_scope.zero())) return r;
    n_scope = 0;
    if ((r = soft.zero())) return r;
    if (fll_st.p && (r = fll_st.zero())) return r;
    if (fam == F_QPSK || fam == F_BPSK || fsk4_disc) {
        std::vector<QpskState> qs(cfg.batch);
        for (auto& q : qs) { std::memset(&q, 0, sizeof q); q.gain = 1.0f; q.avg = q.inst = (float)sps_eff;
                             if (fam == F_BPSK) q.mu = 0.5f; }   // clock_recovery_mm_cc(mu = 0.5), gr_demod_bpsk.cpp:58-60
        if (hipMemcpy(qp_st.p, qs.data(), qs.size() * sizeof(QpskState), hipMemcpyHostToDevice) != hipSuccess) return QRL_ERR_HIP;
    }
    if (dmo_st.p) {
        std::vector<DmoState> ds(cfg.batch);
        for (auto& x : ds) { std::memset(&x, 0, sizeof x); x.endPtr = 9999; }
        if (hipMemcpy(dmo_st.p, ds.data(), ds.size() * sizeof(DmoState), hipMemcpyHostToDevice) != hipSuccess) return QRL_ERR_HIP;
    }
    std::vector<SymSyncState> ss(cfg.batch);
    for (auto& s : ss) { std::memset(&s, 0, sizeof s); s.avg = s.inst = (float)sps_eff; }
    if (hipMemcpy(ss_st.p, ss.data(), ss.size() * sizeof(SymSyncState), hipMemcpyHostToDevice) != hipSuccess) return QRL_ERR_HIP;
    std::vector<FecState> fs((size_t)cfg.batch * 2);
    for (auto& f : fs) { f.consumed = 0; f.start_state = 0; f.last_bits = 0xFE; }  // descrambler seed 0x7F, newest bit first
    if (hipMemcpy(fec_st.p, fs.data(), fs.size() * sizeof(FecState), hipMemcpyHostToDevice) != hipSuccess) return QRL_ERR_HIP;
    if (qp_snap.p && (r = qp_snap.zero())) return r;
    if (fam == F_DSSS) {
        for (auto* b : {&ds_ra, &ds_rb, &ds_rc, &ds_rd, &ds_sym}) if ((r = b->zero())) return r;
        std::vector<DsssState> ds(cfg.batch);
        for (auto& x : ds) { x.phase = 0.f; x.freq = 0.f; x.gain = 10.0f; x.pad = 0.f; }   // agc2_cc(0.1, 0.1, 1, 10), gr_demod_dsss.cpp:61
        if (hipMemcpy(ds_st.p, ds.data(), ds.size() * sizeof(DsssState), hipMemcpyHostToDevice) != hipSuccess) return QRL_ERR_HIP;
        std::vector<DsssTailState> dt(cfg.batch);
        for (auto& x : dt) { std::memset(&x, 0, sizeof x); x.mu = 0.5f; x.omega = 1.0f; }   // clock_recovery_mm_cc(1, ., 0.5, ., .), :69-70
        if (hipMemcpy(ds_tail.p, dt.data(), dt.size() * sizeof(DsssTailState), hipMemcpyHostToDevice) != hipSuccess) return QRL_ERR_HIP;
        n5 = nsy = 0;
    }
    if (fam == F_ANALOG) {
        for (auto* b : {&an_f1, &an_f2, &an_f3, &an_f4}) if (b->p && (r = b->zero())) return r;
        if (an_cs.p) {   // squelch_base: muted, envelope 0 (ramp 160); the Goertzel filters empty
            std::vector<CtcssState> cs(cfg.batch);
            for (auto& x : cs) { std::memset(&x, 0, sizeof x); x.mute = 1; x.env = 0.0; }
            if (hipMemcpy(an_cs.p, cs.data(), cs.size() * sizeof(CtcssState), hipMemcpyHostToDevice) != hipSuccess) return QRL_ERR_HIP;
        }
        if (an_c1.p && (r = an_c1.zero())) return r;
        std::vector<AnState> as(cfg.batch);
        for (auto& x : as) { std::memset(&x, 0, sizeof x); x.env = an_ramp ? 0.0f : 1.0f; x.gain = 1.0f; }   // agc2_ff(0.1, 0.1, 1, 1), gr_demod_am.cpp:48
        if (hipMemcpy(an_st.p, as.data(), as.size() * sizeof(AnState), hipMemcpyHostToDevice) != hipSuccess) return QRL_ERR_HIP;
    }
    q_valid[0] = q_valid[1] = false; tail2_valid[0] = tail2_valid[1] = false; tail_pending = false; call_no = 0;
    fe_valid[0] = fe_valid[1] = false; pre_pending = false;
    fec_deferred = false;
    n_in = n1 = n2 = 0;
    rot_acc = 0; rot_nbase = 0; hist_flip = false;
    return QRL_OK;
}

int qrl_demod::build()
{
    int r;
    const int sps = cfg.sps, samp_rate = cfg.samp_rate, fw = cfg.filter_width;
    if (fam == F_2FSK) {
        if (sps == 10)     { target = 20000; sps_eff = sps;     decim = 50; interp = 1; }
        else if (sps >= 5) { target = 40000; sps_eff = sps * 2; decim = 25; interp = 1; }
        else if (sps == 1) { target = 80000; sps_eff = 4;       decim = 25; interp = 2; }
        else return fail(QRL_ERR_ARG, "2fsk: unsupported sps");
    } else if (fam == F_GMSK) {
        if (sps == 10)     { target = 20000; sps_eff = sps;     decim = 50; interp = 1; }
        else if (sps == 5) { target = 40000; sps_eff = sps * 2; decim = 25; interp = 1; }
        else if (sps == 1) { target = 80000; sps_eff = 4;       decim = 25; interp = 2; }
        else return fail(QRL_ERR_ARG, "gmsk: unsupported sps");
    } else if (fam == F_DMR) {
        // gr_demod_dmr.cpp:36-58, gr_demod_m17.cpp:38-58: 3/125 resampler to 24 ksps, 5 samples per symbol
        target = 24000; sps_eff = 5; decim = 125; interp = 3; branches = 1;
    } else if (fam == F_4FSK) {
        // gr_demod_4fsk.cpp:45-82 (FM branch only; the non-FM discriminator bank of 4FSK2K is not built)
        fsk4_disc = !cfg.fm;   // ModemType4FSK2K: four band-pass magnitudes -> gr_4fsk_discriminator -> symbol_sync_cc (:110-127,165-181)
        if (fsk4_disc && sps == 2) return fail(QRL_ERR_ARG, "4fsk: the sps = 2 geometry exists as FM variant only (gr_demod_4fsk.cpp:78-85 sets no rs/bw)");
        if (sps == 1)       { target = 80000;  sps_eff = 8;  decim = 25;  interp = 2; }
        else if (sps == 5)  { target = 20000;  sps_eff = 10; decim = 50;  interp = 1; }
        else if (sps == 10) { target = 10000;  sps_eff = 10; decim = 100; interp = 1; }
        else if (sps == 2)  { target = 500000; sps_eff = 5;  decim = 2;   interp = 1; }
        else return fail(QRL_ERR_ARG, "4fsk: unsupported sps");
        branches = 1;
    } else if (fam == F_DSSS) {
        // gr_demod_dsss.cpp:37-59: 1:50 to 20 ksps (this stage), then 13/50 to 5 200 samples/s (dsss_stages); sps = samples per chip
        if (sps != 25) return fail(QRL_ERR_ARG, "dsss: sps must be 25 (make_gr_demod_dsss(25, ...), gr_demod_base.cpp:218)");
        target = 20000; sps_eff = 10; decim = 50; interp = 1;
    } else if (fam == F_ANALOG) {
        // gr_demod_nbfm.cpp:39,50 / gr_demod_am.cpp:36,44: 1:50 to 20 ksps; gr_demod_wbfm.cpp:37,49: 1:5 to 200 ksps (sps is unused there)
        // gr_demod_ssb.cpp:35,41-43: 1:sps (125) to 8 ksps
        target = an_kind == 2 ? 200000 : an_kind == 3 ? 8000 : 20000; decim = an_kind == 2 ? 5 : an_kind == 3 ? 125 : 50; interp = 1; sps_eff = 10; branches = 1;
        if (an_kind == 3 && sps != 125) return fail(QRL_ERR_ARG, "ssb: sps must be 125 (make_gr_demod_ssb(125, ...), gr_demod_base.cpp:226-227)");
    } else if (fam == F_BPSK) {
        // gr_demod_bpsk.cpp:40-52: 1:50 to 20 ksps, sps samples per symbol
        if (sps != 10 && sps != 5) return fail(QRL_ERR_ARG, "bpsk: sps must be 10 (BPSK1K) or 5 (BPSK2K)");
        target = 20000; sps_eff = sps; decim = 50; interp = 1;
    } else {
        // gr_demod_qpsk.cpp:39-60: sps <= 4 (QPSK250K / video: 1:2, no FLL), 4 < sps < 125 (QPSK20K: 1:25 to 40 ksps),
        // sps >= 125 (QPSK2K: 1:100 to 10 ksps); the last two run fll_band_edge_cc in front of the shaping filter (:130-138)
        if (sps < 2) return fail(QRL_ERR_ARG, "qpsk: sps must be >= 2");
        if (sps > 4 && sps < 125) { target = 40000; sps_eff = sps * 4 / 25; decim = 25; }
        else if (sps >= 125)      { target = 10000; sps_eff = sps / 25;     decim = 100; }
        else                      { target = 500000; sps_eff = sps;         decim = 2; }
        if (sps_eff < 2 || sps_eff > 10) return fail(QRL_ERR_ARG, "qpsk: unsupported samples per symbol");
        interp = 1; branches = 1; qpsk_fll = sps > 4;
    }
    fm = cfg.fm != 0;
    const int B = cfg.batch;
    const size_t maxn = cfg.max_chunk;

    // --- front end (gr_demod_base.cpp:1317-1340)
    fe_decim = 1;
    if (cfg.device_samp_rate >= 2000000) {
        fe_decim = cfg.device_samp_rate / 1000000;
        if ((r = fe.plan(low_pass(1, cfg.device_samp_rate, 480000, 100000, WIN_BLACKMAN_HARRIS), fe_decim))) return fail(r, "front-end plan");
        if ((r = fe.alloc_edge(cfg.batch))) return fail(r, "front-end edge scratch");
    }
    rot_inc = phase_inc_to_turn(2 * M_PI * -cfg.carrier_offset_hz / cfg.device_samp_rate);
    if ((r = upload_rot_table())) return r;

    // --- per-mode first resampler (gr_demod_2fsk.cpp:82-88, gr_demod_gmsk.cpp:80-83)
    const std::vector<float> rtaps = fam == F_DMR && m17
        ? low_pass(3, (double)samp_rate * 3, target / 2, target / 2, WIN_BLACKMAN_HARRIS)                    // gr_demod_m17.cpp:55-58
        : fam == F_DMR
        ? low_pass_2(3, (double)samp_rate * 3, 5000, 2000, 60, WIN_BLACKMAN_HARRIS)                          // gr_demod_dmr.cpp:55-58
        : fam == F_QPSK
        ? low_pass_2(interp, (double)interp * samp_rate, target / 2, target / 10, 60, WIN_BLACKMAN_HARRIS)   // gr_demod_qpsk.cpp:92-96
        : low_pass(interp, (double)interp * samp_rate, target / 2, target / 2, WIN_BLACKMAN_HARRIS);
    if (interp == 1) { if ((r = first.plan(rtaps, decim))) return fail(r, "resampler plan"); if (!fe.used && (r = first.alloc_edge(cfg.batch))) return fail(r, "resampler edge scratch"); }
    else {
        rs_Jp = ((int)rtaps.size() + interp - 1) / interp;
        if ((r = rs_taps.upload(resamp_layout(rtaps, interp, rs_Jp)))) return r;
    }

    // QPSK-250k class (gr_demod_qpsk.cpp:92-103 with sps <= 4): 1:2 resampler and the RRC behind it run as ONE kernel
    std::vector<float> d2f_rrc;
    if (fam == F_QPSK && !qpsk_fll && interp == 1) {
        d2f_rrc = root_raised_cosine(sps_eff, sps_eff, 1, 0.35, 11 * sps_eff);
        d2f_capable = d2f = dec2_fir_supported((int)rtaps.size(), decim, (int)d2f_rrc.size());
        if (d2f && (r = d2f_taps.upload(dec2_fir_table(rtaps, d2f_rrc)))) return r;
    }
    const uint32_t first_look = interp == 1 ? std::max<uint32_t>(first.lookback(), d2f ? dec2_fir_lookback() : 0u) : 0u;
    // --- the scope tap's 1:10 decimator on the 1 Msps signal (planned here so that the history below covers it; its ring is allocated on first use)
    {
        const int sr = cfg.time_domain_samp_rate;
        if (sr < 0 || sr > 500000 || cfg.time_domain_filter_width < 0 || cfg.time_domain_filter_width > 500000) return fail(QRL_ERR_ARG, "time_domain_samp_rate / filter_width out of range");
        scope_D = sr > 0 ? 1000000 / sr : 10;
        if (sr > 0 && sr / 2 - sr / 8 <= 0) return fail(QRL_ERR_ARG, "time_domain_samp_rate too small");
        const std::vector<float> h = cfg.time_domain_filter_width > 0 ? low_pass(1, 1000000, cfg.time_domain_filter_width, cfg.time_domain_filter_width, WIN_HAMMING)   // gr_demod_base.cpp:1292-1301
                                   : sr > 0 ? low_pass(1, 1000000, sr / 2 - sr / 8, sr / 4, WIN_HAMMING)                                                                  // :1249-1290
                                            : low_pass(1, 1000000, 50000, 25000, WIN_HAMMING);                                                                            // :62-63
        if (h.size() > 4096) return fail(QRL_ERR_ARG, "time-domain filter too long (> 4096 taps)");
        if ((r = scope.plan(h, scope_D))) return fail(r, "scope plan");
    }
    if (!fe.used && (r = scope.alloc_edge(cfg.batch))) return fail(r, "scope edge scratch");
    // --- history of the caller's IQ kept by whichever stage reads it
    if (fe.used) hist_len = fe.lookback();
    else if (interp == 1) hist_len = std::max(first_look, scope.lookback());
    else hist_len = std::max((uint32_t)(rs_Jp + decim + 2), scope.lookback());
    if ((r = hist_a.alloc((size_t)B * hist_len)) || (r = hist_b.alloc((size_t)B * hist_len))) return r;

    // --- rings
    const size_t max1 = fe.used ? maxn / fe_decim + 2 : 0;           // 1 Msps items per call
    const size_t in2 = fe.used ? max1 : maxn;                        // items entering the mode resampler per call
    const size_t max2 = in2 * interp / decim + 2;                    // target-rate items per call
    if (fe.used) {
        const size_t look = std::max<size_t>(interp == 1 ? first_look : (size_t)(rs_Jp + decim + 2), scope.lookback());
        s1_mask = pow2_at_least(max1 + look + 64) - 1;
        if ((r = s1.alloc((size_t)B * (s1_mask + 1)))) return r;
    }
    // default: only the 2FSK family, whose FLL + discriminator kernels are a third of a call (measured, C1: 15.2 -> 12.9 ms per
    // step); for the light GMSK / 4FSK tails the extra stream hand-over costs more than it hides (C2: 2.86 -> 3.05 ms).
    // Measured on C1 (round 3, same box): 8.79 instead of 9.49 ms per step, while the front-end kernel, sharing the GPU, stretches
    // from 6.33 to 8.04 ms -- the recursion kernels are only placed once front-end workgroups drain, and a 68 KB FLL workgroup then
    // takes the place of two of them.
    overlap_capable = fam == F_2FSK;
    grouped_capable = fam == F_QPSK && !fsk4_disc;   // the chain whose recursion is k_qpsk_pipe4
    {
        int dev = 0, cus = 256; hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) cus = pr.multiProcessorCount;
        grouped = grouped_capable && 2 * ((B + 63) / 64) >= cus;   // (below that the recursion's workgroups leave CUs free: the front end of the next call belongs beside it -- C3)
    }
    overlap = overlap_capable;   // round 3: ON by default for the 2FSK family (same-box A/B on C1: 8.79 against 9.49 ms per step); qrl_demod_set_option(QRL_OPT_OVERLAP, 0) gives the serial order
    s2_mask = pow2_at_least((overlap_capable || loops_family() ? 2 : 1) * max2 + (fam == F_DMR ? 2048 : fam == F_ANALOG ? 4096 : 1024)) - 1;   // DMR: the DMO slicer looks back 1440 samples   // history needs: <= 501 taps downstream; overlapped mode: two calls
    const size_t ring2 = (size_t)B * (s2_mask + 1);
    if ((r = s2.alloc(ring2)) || (r = s2f.alloc(ring2)) || (r = s2d.alloc(ring2)) || (r = s3.alloc(ring2))) return r;
    if ((fam == F_2FSK || fam == F_BPSK || (fam == F_QPSK && qpsk_fll)) && (r = s2l.alloc(ring2))) return r;
    const size_t maxsym = max2 / (size_t)(sps_eff > 1 ? sps_eff - 1 : 1) + 8;
    soft_mask = pow2_at_least((loops_family() ? 2 : 1) * (fam == F_QPSK || fam == F_4FSK ? 2 : 1) * maxsym + 512) - 1;   // loops families: two calls (decoder of call k beside the recursion of call k + 1)
    if ((r = soft.alloc((size_t)B * (soft_mask + 1)))) return r;

    // --- decimated-rate filters
    {
        const std::vector<float> f = fam == F_QPSK
            ? root_raised_cosine(sps_eff, sps_eff, 1, 0.35, 11 * sps_eff)      // _shaping_filter, gr_demod_qpsk.cpp:100-103
            : fam == F_BPSK
            ? root_raised_cosine(sps_eff, sps_eff, 1, 0.35, 15 * sps_eff)      // _shaping_filter, gr_demod_bpsk.cpp:64-66
            : fam == F_4FSK
            ? low_pass(1, target, fw, fw / 2, WIN_BLACKMAN_HARRIS)             // _filter, gr_demod_4fsk.cpp:108-109
            : fam == F_ANALOG
            ? (an_kind == 2 ? low_pass_2(1, target, fw, 600, 90, WIN_BLACKMAN_HARRIS)      // gr_demod_wbfm.cpp:52-53
                            : low_pass_2(1, target, fw, 3500, 60, WIN_BLACKMAN_HARRIS))    // gr_demod_nbfm.cpp:53-54 (AM: an_filt_c)
            : low_pass(1, target, fw, fw, WIN_BLACKMAN_HARRIS);
        filt_nt = (int)f.size();
        if ((r = filt_taps.upload(f))) return r;
    }
    if ((r = atan_tab.upload(atan_table())) || (r = mmse_tab.upload(mmse_table()))) return r;
    if (fam == F_2FSK) {
        std::vector<std::complex<float>> lo, up;
        fll_band_edge_taps((float)sps_eff, 0.1f, 16, lo, up);
        if ((r = fll_lo.upload(to_f2(lo))) || (r = fll_up.upload(to_f2(up)))) return r;
        control_loop_gains((float)(24 * M_PI / 100), fll_alpha, fll_beta);
        fll_maxf = (float)(2 * M_PI * (2.0 / sps_eff));
        if ((r = fll_st.alloc(B))) return r;
        if (fm) {
            int nfilts = (sps == 1 ? 125 : 35) * sps_eff;
            if ((nfilts % 2) == 0) nfilts += 1;
            const std::vector<float> rrc = root_raised_cosine(1, target, target / sps_eff, 0.2, nfilts);
            symf_nt = (int)rrc.size();
            if ((r = symf_taps.upload(rrc))) return r;
            demod_gain = (float)(sps_eff / (1 * M_PI / 2));
        } else {
            const auto up2 = complex_band_pass(1, target, -fw, 0, fw, WIN_BLACKMAN_HARRIS);
            const auto lo2 = complex_band_pass(1, target, 0, fw, fw, WIN_BLACKMAN_HARRIS);
            disc_nt = (int)up2.size();
            // the discriminator kernels use the pair as what it is -- lower = conj(upper), bit for bit (same prototype, cos(-x) = cos(x),
            // sin(-x) = -sin(x)) -- and run the shared real-tap chains once (oracle orc_fir_ccc_conj_pair)
            for (size_t k = 0; k < up2.size(); ++k) {
                const float ur = up2[k].real(), ui = -up2[k].imag(), lr = lo2[k].real(), li = lo2[k].imag();
                if (std::memcmp(&ur, &lr, sizeof ur) || std::memcmp(&ui, &li, sizeof ui)) return fail(QRL_ERR_ARG, "2FSK discriminator filters are not a conjugate pair");
            }
            if ((r = disc_up.upload(to_f2(up2))) || (r = disc_lo.upload(to_f2(lo2)))) return r;
            const std::vector<float> sf = low_pass(1.0, target, target / sps_eff, target / sps_eff, WIN_HAMMING);
            symf_nt = (int)sf.size();
            if ((r = symf_taps.upload(sf))) return r;
            // zero-padded copies for the fused kernel (4 A + 1 taps, tables of 4 (A + 1) entries)
            auto padf = [](std::vector<float> v) { v.resize((size_t)fsk2_ff_padded((int)v.size()) + 3, 0.0f); return v; };
            auto padc = [](std::vector<float2> v) { v.resize((size_t)fsk2_ff_padded((int)v.size()) + 3, make_float2(0.f, 0.f)); return v; };
            const std::vector<float> ftaps = low_pass(1, target, fw, fw, WIN_BLACKMAN_HARRIS);
            if ((r = ff_tf.upload(padf(ftaps))) || (r = ff_ts.upload(padf(sf))) || (r = ff_up.upload(padc(to_f2(up2)))) ||
                (r = ff_lo.upload(padc(to_f2(lo2))))) return r;
        }
        const float symbol_rate = (float)target / (float)sps_eff;
        const float dev = 200.0f / symbol_rate;
        clock_loop_gains((float)(2 * M_PI / (symbol_rate / 10)), 1.0f, 0.2869f, ss_alpha, ss_beta);
        ss_maxp = (float)sps_eff + dev; ss_minp = (float)sps_eff - dev;
    } else if (fam == F_DMR && m17) {
        const std::vector<float> rrc = root_raised_cosine(1.5, target, target / sps_eff, 0.5, 50 * sps_eff); // gr_demod_m17.cpp:64-67
        symf_nt = (int)rrc.size();
        if ((r = symf_taps.upload(rrc))) return r;
        demod_gain = (float)(sps_eff / M_PI);                                                                // :63
        const float symbol_rate = (float)target / (float)sps_eff;
        clock_loop_gains((float)(2 * M_PI / (symbol_rate / 50)), 1.0f, 0.2869f, ss_alpha, ss_beta);          // :70-71
        ss_maxp = (float)sps_eff + 500.0f / symbol_rate; ss_minp = (float)sps_eff - 500.0f / symbol_rate;
    } else if (fam == F_DMR) {
        const std::vector<float> rrc = root_raised_cosine(1, target, target / sps_eff, 0.2, 25 * sps_eff);   // gr_demod_dmr.cpp:62-66
        symf_nt = (int)rrc.size();
        if ((r = symf_taps.upload(rrc))) return r;
        demod_gain = (float)(target / (M_PI / 2 * (float)(target / sps_eff)));                               // :72
        clock_loop_gains((float)(2 * M_PI / 100.0f), 1.0f, 0.2869f, ss_alpha, ss_beta);                      // :70-71
        ss_maxp = (float)sps_eff + 0.06f; ss_minp = (float)sps_eff - 0.06f;
    } else if (fam == F_4FSK && fsk4_disc) {
        const int rs = sps == 1 ? 10000 : sps == 5 ? 2000 : 1000, bw = sps == 10 ? 2000 : 4000;           // gr_demod_4fsk.cpp:45-76
        const int lo_[4] = {-fw, -fw + rs, 0, fw - rs}, hi_[4] = {-fw + rs, 0, fw - rs, fw};                  // _filter1..4, :112-119
        std::vector<float2> bt;
        for (int q = 0; q < 4; ++q) {
            const auto t4 = complex_band_pass(1, target, lo_[q], hi_[q], bw, WIN_BLACKMAN_HARRIS);
            disc4_nt = (int)t4.size();
            const auto f2 = to_f2(t4);
            bt.insert(bt.end(), f2.begin(), f2.end());
        }
        if ((r = disc4_taps.upload(bt))) return r;
        const std::vector<float> st4 = low_pass(1.0, target, target / sps_eff, target / sps_eff / 20, WIN_BLACKMAN_HARRIS);   // :103-105
        sym4_nt = (int)st4.size();
        if ((r = sym4_taps.upload(st4)) || (r = s2g.alloc(ring2)) || (r = s2l.alloc(ring2))) return r;
        if ((r = tanh_tab.upload(tanh_table())) || (r = qp_st.alloc(B))) return r;
        clock_loop_gains((float)(2 * M_PI / 200.0f), 1.0f, 0.2869f, ss_alpha, ss_beta);                      // :138-140
        ss_maxp = (float)sps_eff + 0.05f; ss_minp = (float)sps_eff - 0.05f;
    } else if (fam == F_4FSK) {
        int nfilts = (sps == 1 ? 32 : sps == 2 ? 50 : 25) * sps_eff;                                         // gr_demod_4fsk.cpp:45-84
        if ((nfilts % 2) == 0) nfilts += 1;
        const std::vector<float> rrc = root_raised_cosine(1.5, target, target / sps_eff, 0.2, nfilts);       // :130-133
        symf_nt = (int)rrc.size();
        if ((r = symf_taps.upload(rrc))) return r;
        demod_gain = (float)(sps_eff / (1 * M_PI));                                                          // :129
        clock_loop_gains((float)(2 * M_PI / 200.0f), 1.0f, 0.2869f, ss_alpha, ss_beta);                      // :135-137
        ss_maxp = (float)sps_eff + 0.05f; ss_minp = (float)sps_eff - 0.05f;
    } else if (fam == F_BPSK) {
        std::vector<std::complex<float>> lo, up;
        fll_band_edge_taps((float)sps_eff, 0.35f, 32, lo, up);                                               // gr_demod_bpsk.cpp:63
        if ((r = fll_lo.upload(to_f2(lo))) || (r = fll_up.upload(to_f2(up)))) return r;
        control_loop_gains((float)(8 * M_PI / 100), fll_alpha, fll_beta);
        fll_maxf = (float)(2 * M_PI * (2.0 / sps_eff));
        if ((r = fll_st.alloc(B))) return r;
        if ((r = tanh_tab.upload(tanh_table())) || (r = qp_st.alloc(B))) return r;
        control_loop_gains((float)(2 * M_PI / 200), c2_alpha, c2_beta);                                      // _costas_loop, :61
    } else if (fam == F_QPSK) {
        if ((r = tanh_tab.upload(tanh_table())) || (r = qp_st.alloc(B))) return r;
        control_loop_gains((float)(M_PI / 200 / sps_eff), c1_alpha, c1_beta);     // _costas_pll, gr_demod_qpsk.cpp:110
        control_loop_gains((float)(qpsk_fll ? M_PI / 200 : M_PI / 400), c2_alpha, c2_beta);   // _costas_loop, :44,67,112
        if (qpsk_fll) {   // _fll = fll_band_edge_cc(sps, 0.35, 32, 2 pi / 100), :98-99
            std::vector<std::complex<float>> lo, up;
            fll_band_edge_taps((float)sps_eff, 0.35f, 32, lo, up);
            if ((r = fll_lo.upload(to_f2(lo))) || (r = fll_up.upload(to_f2(up)))) return r;
            control_loop_gains((float)(2 * M_PI / 100), fll_alpha, fll_beta);
            fll_maxf = (float)(2 * M_PI * (2.0 / sps_eff));
            if ((r = fll_st.alloc(B))) return r;
        }
        const float symbol_rate = (float)target / (float)sps_eff;
        const float dev = 200.0f / symbol_rate;
        clock_loop_gains((float)(2 * M_PI / (symbol_rate / 10)), 1.0f, 0.2869f, ss_alpha, ss_beta);
        ss_maxp = (float)sps_eff + dev; ss_minp = (float)sps_eff - dev;
        const float ang = (float)(-3 * M_PI / 4);
        qp_rot = make_float2((float)std::cos((double)ang), (float)std::sin((double)ang));
    } else {
        const std::vector<float> sf = low_pass(1, target, target / sps_eff, target / sps_eff, WIN_HAMMING);
        symf_nt = (int)sf.size();
        if ((r = symf_taps.upload(sf))) return r;
        demod_gain = (float)(sps_eff / (M_PI / 2));
        clock_loop_gains((float)(2 * M_PI / 200.0f), 1.0f, 0.2869f, ss_alpha, ss_beta);
        ss_maxp = (float)sps_eff + 0.05f; ss_minp = (float)sps_eff - 0.05f;
    }
    if ((r = ss_st.alloc(B)) || (r = fec_st.alloc((size_t)B * 2)) || (r = counts_scratch.alloc((size_t)B * 4))) return r;
    if (loops_family() && (r = qp_snap.alloc((size_t)B * 2))) return r;
    if (fam == F_ANALOG) {
        std::vector<float> rt, ft;
        double a[2], b[2];
        if (an_kind == 0) {          // gr_demod_nbfm.cpp:43-64
            an_ramp = 320; an_I = 2; an_D = 5;
            rt = low_pass_2(2, 2 * target, 3600, 250, 60, WIN_BLACKMAN_HARRIS);
            ft = low_pass_2(1, 8000, 3500, 200, 35, WIN_BLACKMAN_HARRIS);
            an_gain = (float)(target / (4 * M_PI * fw));
            deemph_taps(target, 50e-6, a, b);
            an_de_ff[0] = b[0]; an_de_ff[1] = b[1]; an_de_fb1 = -a[1];            // iir_filter_ffd(btaps, ataps, oldstyle = false)
        } else if (an_kind == 1) {   // gr_demod_am.cpp:40-61
            an_ramp = 0; an_I = 2; an_D = 5;
            rt = low_pass(2, 2 * target, 3600, 600, WIN_BLACKMAN_HARRIS);
            ft = low_pass(1, 8000, 3600, 300, WIN_BLACKMAN_HARRIS);
            const auto fc = complex_band_pass_2(1, target, -fw, fw, 200, 90, WIN_BLACKMAN_HARRIS);
            an_nfc = (int)fc.size();
            if ((r = an_filt_c.upload(to_f2(fc)))) return r;
            an_ff[0] = 1; an_ff[1] = -1; an_fb1 = 0.9999;                          // iir_filter_ffd({1, -1}, {0, 0.9999}), old style
        } else if (an_kind == 3) {   // gr_demod_ssb.cpp:41-58
            an_ramp = 0; an_I = 0; an_D = 1;                                       // I = 0: the stretcher's chunked output count
            ft = band_pass_2(1, target, 200, fw, 200, 90, WIN_BLACKMAN_HARRIS);
            const auto fc = an_lsb ? complex_band_pass_2(1, target, -fw, -200, 200, 90, WIN_BLACKMAN_HARRIS)
                                   : complex_band_pass_2(1, target, 200, fw, 200, 90, WIN_BLACKMAN_HARRIS);
            an_nfc = (int)fc.size();
            if ((r = an_filt_c.upload(to_f2(fc)))) return r;
        } else {                     // gr_demod_wbfm.cpp:41-57
            an_ramp = 0; an_I = 1; an_D = 25;
            rt = low_pass(1, target, 4000, 2000, WIN_BLACKMAN_HARRIS);
            an_gain = (float)(target / (2 * M_PI * fw));
            deemph_taps(8000, 50e-6, a, b);
            an_ff[0] = b[0]; an_ff[1] = b[1]; an_fb1 = -a[1];
        }
        an_nr = (int)rt.size(); an_nf = (int)ft.size();
        if ((an_nr && (r = an_rtaps.upload(rt))) || (an_nf && (r = an_ftaps.upload(ft))) || (r = an_env.upload(squelch_envelope(an_ramp)))) return r;
        an_m1 = pow2_at_least(max2 + 2048) - 1;                                  // the audio resampler looks <= 419 gated items back, the stretcher <= 1025
        an_m2 = an_kind == 3 ? an_m1 : pow2_at_least(max2 * an_I / an_D + 512) - 1;
        if (an_kind == 3) { if ((r = an_c1.alloc((size_t)B * (an_m1 + 1)))) return r; }
        else if ((r = an_f1.alloc((size_t)B * (an_m1 + 1)))) return r;
        if ((r = an_f2.alloc((size_t)B * (an_m2 + 1))) || ((an_kind == 0 || an_kind == 1) && (r = an_f3.alloc((size_t)B * (an_m2 + 1)))) || (r = an_st.alloc(B))) return r;
    }
    if (fam == F_DSSS) {
        const std::vector<float> ti = low_pass(1, target, 2600, 2600, WIN_BLACKMAN_HARRIS);              // _resampler_if (13, 50), gr_demod_dsss.cpp:57-59
        ds_Jp = ((int)ti.size() + 12) / 13;
        const std::vector<float> tf = low_pass(1, 5200, fw, 1200, WIN_BLACKMAN_HARRIS);                 // _filter, :62-63
        ds_nf = (int)tf.size();
        if ((r = ds_rs.upload(resamp_layout(ti, 13, ds_Jp))) || (r = ds_filt.upload(tf)) || (r = ds_mf.upload(dsss_matched_filter(sps)))) return r;
        if (!tanh_tab.p && (r = tanh_tab.upload(tanh_table()))) return r;
        control_loop_gains((float)(M_PI / 200), ds_a1, ds_b1);                                          // _costas_freq, :64
        control_loop_gains((float)(2 * M_PI / 100), ds_a2, ds_b2);                                      // _costas_loop, :63
        const size_t max5 = max2 * 13 / 50 + 2;
        ds_mask = pow2_at_least(max5 + 2048) - 1;                                                       // the matched filter looks 2 x 325 + 600 items back
        ds_sym_mask = pow2_at_least(max5 / 325 + 64) - 1;
        const size_t r5 = (size_t)B * (ds_mask + 1);
        if ((r = ds_ra.alloc(r5)) || (r = ds_rb.alloc(r5)) || (r = ds_rc.alloc(r5)) || (r = ds_rd.alloc(r5)) ||
            (r = ds_sym.alloc((size_t)B * (ds_sym_mask + 1))) || (r = ds_st.alloc(B)) || (r = ds_tail.alloc(B))) return r;
    }
    return init_state();
}

int qrl_demod::process(const float* iq, size_t stride, size_t n, const qrl_demod_out* out)
{
    if (n > cfg.max_chunk) return fail(QRL_ERR_TOO_BIG, "n exceeds max_chunk");
    if ((reinterpret_cast<uintptr_t>(iq) & 15u) || (stride & 1u)) return fail(QRL_ERR_ARG, "iq must be 16-byte aligned, stride even");
    const int B = cfg.batch;
    uint32_t* counts = (out && out->counts) ? out->counts : counts_scratch.p;
    hipStream_t cs = overlap ? tail : stream;   // stream of stage C (decimated-rate feed-forward kernels)
    const int slot = (int)(call_no & 1);
    if (overlap) {
        // ring s2 is about to be overwritten two calls behind: the tail of call k - 2 must be through
        if (tail2_valid[slot]) HIPCHK(hipStreamWaitEvent(stream, ev_tail2[slot], 0));
    } else {
        HIPCHK(hipMemsetAsync(counts, 0, (size_t)B * 4 * sizeof(uint32_t), stream));
    }
    // loops families: the rings the recursion reads hold two calls; call k - 2's recursion must be through before they are rewritten
    if (loops_family() && q_valid[slot]) HIPCHK(hipStreamWaitEvent(stream, ev_q[slot], 0));
    if (grouped && q_valid[slot ^ 1]) HIPCHK(hipStreamWaitEvent(stream, ev_q[slot ^ 1], 0));   // grouped order: this front end behind the recursion of the call before
    if (pre_pending) { HIPCHK(hipStreamWaitEvent(stream, ev_pre, 0)); pre_pending = false; }   // k_hist of the call before (helper stream)
    const bool use_pre = pre && input_resident;
    const float2* in = reinterpret_cast<const float2*>(iq);
    const float2* hist_old = hist_flip ? hist_b.p : hist_a.p;
    float2* hist_new = hist_flip ? hist_a.p : hist_b.p;

    const uint64_t n_in0 = n_in, n_in1 = n_in + n;
    uint64_t n1_0 = n1, n1_1 = n1;
    RingC r1{s1.p, s1_mask}, r2{s2.p, s2_mask}, r2l{s2l.p, s2_mask}, r2f{s2f.p, s2_mask};
    RingF r2d{s2d.p, s2_mask}, r3{s3.p, s2_mask};

    // the HBM-facing kernel is the first launch that reads the caller's IQ
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (profiling) {
        HIPCHK(hipEventCreate(&ev0)); HIPCHK(hipEventCreate(&ev1));
        HIPCHK(hipEventRecord(ev0, stream));
    }
    // ---- stage A: gr_demod_base front end
    if (fe.used) {
        n1_1 = decim_count(n_in1, 1, fe_decim);
        DecimParams p{};
        p.in = in; p.in_stride = stride; p.n0 = n_in0; p.n = (uint32_t)n;
        p.hist = hist_old; p.hist_len = hist_len;
        p.out = r1; p.m0 = n1_0; p.m_count = (uint32_t)(n1_1 - n1_0);
        p.taps = fe.taps.p; p.D = fe.D; p.Jpad = fe.Jpad;
        p.rot_enable = 1; p.rot_acc = rot_acc; p.rot_inc = rot_inc; p.rot_nbase = rot_nbase; p.rot_lo = rot_lo.p;
        if (use_pre) { p.pre_stream = pre; p.pre_event = ev_pre; }
        if (fe.launch(p, B, stream, use_pre ? slot : 0)) return fail(QRL_ERR_HIP, "front-end launch: hipFuncSetAttribute failed");
    }
    if (profiling && fe.used) { HIPCHK(hipEventRecord(ev1, stream)); prof_events.emplace_back(ev0, ev1); }
    // ---- stage B: per-mode resampler
    const uint64_t src0 = fe.used ? n1_0 : n_in0, src1 = fe.used ? n1_1 : n_in1;
    const uint64_t n2_0 = n2, n2_1 = decim_count(src1, interp, decim);
    if (interp == 1) {
        DecimParams p{};
        if (fe.used) { p.in = nullptr; p.in_ring = r1; }
        else { p.in = in; p.in_stride = stride; p.hist = hist_old; p.hist_len = hist_len;
               p.rot_enable = 1; p.rot_acc = rot_acc; p.rot_inc = rot_inc; p.rot_nbase = rot_nbase; p.rot_lo = rot_lo.p; }
        p.n0 = src0; p.n = (uint32_t)(src1 - src0);
        p.out = r2; p.m0 = n2_0; p.m_count = (uint32_t)(n2_1 - n2_0);
        p.taps = first.taps.p; p.D = first.D; p.Jpad = first.Jpad;
        if (d2f) {   // + _shaping_filter -> port 0 and the filtered ring, in the same kernel
            const bool sd = cfg.enable_side_outputs && out;
            Dec2FirParams f{};
            f.d = p; f.taps = d2f_taps.p; f.out = RingC{s2f.p, s2_mask};
            f.port = sd && out->filtered ? reinterpret_cast<float2*>(out->filtered) : nullptr;
            f.port_cap = sd ? out->filtered_cap : 0;
            f.counts = counts;
            launch_dec2_fir(f, B, stream);
        } else {
            if (use_pre && !fe.used) { p.pre_stream = pre; p.pre_event = ev_pre; }   // (device rate 1 Msps: this stage is the one that reads the caller's IQ)
            if (first.launch(p, B, stream, use_pre && !fe.used ? slot : 0)) return fail(QRL_ERR_HIP, "first-stage launch: hipFuncSetAttribute failed");
        }
    } else {
        ResampParams p{};
        if (fe.used) { p.in = nullptr; p.in_ring = r1; }
        else { p.in = in; p.in_stride = stride; p.hist = hist_old; p.hist_len = hist_len;
               p.rot_enable = 1; p.rot_acc = rot_acc; p.rot_inc = rot_inc; p.rot_nbase = rot_nbase; p.rot_lo = rot_lo.p; }
        p.n0 = src0; p.n = (uint32_t)(src1 - src0);
        p.out = r2; p.q0 = n2_0; p.q_count = (uint32_t)(n2_1 - n2_0);
        p.taps = rs_taps.p; p.I = interp; p.D = decim; p.Jp = rs_Jp;
        if (fam == F_DMR && !m17) {   // port 0 of gr_demod_dmr is the resampler output (gr_demod_dmr.cpp:89)
            const bool sd = cfg.enable_side_outputs && out;
            p.port = sd && out->filtered ? reinterpret_cast<float2*>(out->filtered) : nullptr;
            p.port_cap = sd ? out->filtered_cap : 0;
            p.port_counts = counts;
        }
        launch_resamp(p, B, stream);
    }
    if (profiling && !fe.used) { HIPCHK(hipEventRecord(ev1, stream)); prof_events.emplace_back(ev0, ev1); }
    // ---- time-domain scope tap: the 1 Msps signal behind the front end (the caller's rotated IQ when the device runs at 1 Msps) -> 1:10
    if (scope_out) {
        const uint64_t ns_1 = decim_count(src1, 1, scope_D);
        DecimParams p{};
        if (fe.used) { p.in = nullptr; p.in_ring = r1; }
        else { p.in = in; p.in_stride = stride; p.hist = hist_old; p.hist_len = hist_len;
               p.rot_enable = 1; p.rot_acc = rot_acc; p.rot_inc = rot_inc; p.rot_nbase = rot_nbase; p.rot_lo = rot_lo.p; }
        p.n0 = src0; p.n = (uint32_t)(src1 - src0);
        p.out = RingC{s_scope.p, scope_mask}; p.m0 = n_scope; p.m_count = (uint32_t)(ns_1 - n_scope);
        p.taps = scope.taps.p; p.D = scope.D; p.Jpad = scope.Jpad;
        if (scope.launch(p, B, stream)) return fail(QRL_ERR_HIP, "scope launch: hipFuncSetAttribute failed");
        launch_ring_store(RingC{s_scope.p, scope_mask}, n_scope, (uint32_t)(ns_1 - n_scope), scope_out, scope_cap, scope_counts, B, stream);
        n_scope = ns_1;
    }
    // keep the tail of the caller's IQ (rotated) for the next call
    {
        HistParams h{};
        h.in = in; h.in_stride = stride; h.n0 = n_in0; h.n = (uint32_t)n;
        h.hist_old = hist_old; h.hist_new = hist_new; h.hist_len = hist_len;
        h.rot_enable = 1; h.rot_acc = rot_acc; h.rot_inc = rot_inc; h.rot_nbase = rot_nbase; h.rot_lo = rot_lo.p;
        if (pre) { HIPCHK(hipEventRecord(ev_fe[slot], stream)); fe_valid[slot] = true; }   // everything of this call that reads the history on the handle's stream has been launched
        if (use_pre) {
            if (fe_valid[slot ^ 1]) HIPCHK(hipStreamWaitEvent(pre, ev_fe[slot ^ 1], 0));   // hist_new was the history of the call before
            launch_hist_save(h, B, pre);
            HIPCHK(hipEventRecord(ev_pre, pre));
            pre_pending = true;
        } else launch_hist_save(h, B, stream);
        hist_flip = !hist_flip;
    }
    if (overlap) {
        HIPCHK(hipEventRecord(ev_ff, stream));
        HIPCHK(hipStreamWaitEvent(tail, ev_ff, 0));
        HIPCHK(hipMemsetAsync(counts, 0, (size_t)B * 4 * sizeof(uint32_t), tail));
    }
    // ---- stage C: decimated-rate feed-forward (+ FLL for 2FSK)
    const uint32_t c2 = (uint32_t)(n2_1 - n2_0);
    const bool side = cfg.enable_side_outputs && out;
    if (fam == F_DSSS || fam == F_ANALOG) {
        if (int rr = fam == F_DSSS ? dsss_stages(n2_0, n2_1, out, counts, side) : analog_stages(n2_0, n2_1, out, counts, side)) return rr;
        HIPCHK(hipGetLastError());
        if (take_launch_error()) return QRL_ERR_HIP;
        n_in = n_in1; n1 = n1_1; n2 = n2_1; ++call_no;
        return QRL_OK;
    }
    RingC filt_in = r2;
    if (fam == F_2FSK || fam == F_BPSK || (fam == F_QPSK && qpsk_fll)) {
        FllParams f{};
        f.in = r2; f.out = r2l; f.q0 = n2_0; f.count = c2; f.st = fll_st.p;
        f.lower = fll_lo.p; f.upper = fll_up.p; f.nt = fam == F_2FSK ? 16 : 32; f.alpha = fll_alpha; f.beta = fll_beta; f.max_freq = fll_maxf;
        // (slim single-wave FLL workgroups under an 8-wave-workgroup front end were measured in round 3: the front end alone slows from
        //  6.57 to 7.24 ms with 8-wave workgroups and stretches to 8.4 - 9.1 ms when it shares the SIMDs; 9.47 ms per step against 9.29)
        f.slim = fll_slim ? 1 : 0;
        if (!(QRL_DEV_SKIP & 1)) launch_fll(f, B, cs);
        filt_in = r2l;
    }
    const bool fused_2fsk = fam == F_2FSK && !fm && filt_nt <= 41 && disc_nt <= 41 && symf_nt <= 25;
    if (fam == F_DMR) {
        RingC dem_in = r2;
        if (m17) {   // gr_demod_m17.cpp:60-61,89-90: channel filter behind the resampler, its output is port 0
            FirCcfParams f{};
            f.in = r2; f.out = r2f; f.q0 = n2_0; f.count = c2; f.taps = filt_taps.p; f.nt = filt_nt;
            f.port = side && out->filtered ? reinterpret_cast<float2*>(out->filtered) : nullptr;
            f.port_cap = side ? out->filtered_cap : 0;
            f.counts = counts;
            launch_fir_ccf(f, B, cs);
            dem_in = r2f;
        }
        QuadDemodParams q{}; q.in = dem_in; q.out = r2d; q.q0 = n2_0; q.count = c2; q.gain = demod_gain; q.atan_tab = atan_tab.p;
        launch_quad_demod(q, B, cs);
        if (!overlap && tail_pending) { HIPCHK(hipStreamWaitEvent(stream, ev_tail, 0)); tail_pending = false; }
        FirFffParams f{}; f.in = r2d; f.out = r3; f.q0 = n2_0; f.count = c2; f.taps = symf_taps.p; f.nt = symf_nt;
        launch_fir_fff(f, B, cs);
        if (!overlap) { HIPCHK(hipEventRecord(ev_ff, stream)); HIPCHK(hipStreamWaitEvent(tail, ev_ff, 0)); }
    } else if (fused_2fsk) {
        if (!overlap && tail_pending) { HIPCHK(hipStreamWaitEvent(stream, ev_tail, 0)); tail_pending = false; }
        Fsk2FfParams f{};
        f.in = filt_in; f.out = r3; f.q0 = n2_0; f.count = c2;
        f.tf = ff_tf.p; f.nf = fsk2_ff_padded(filt_nt); f.up = ff_up.p; f.lo = ff_lo.p; f.nb = fsk2_ff_padded(disc_nt);
        f.ts = ff_ts.p; f.ns = fsk2_ff_padded(symf_nt);
        f.port = side && out->filtered ? reinterpret_cast<float2*>(out->filtered) : nullptr;
        f.port_cap = side ? out->filtered_cap : 0;
        f.counts = counts;
        if (!(QRL_DEV_SKIP & 2)) launch_2fsk_ff(f, B, cs);
        if (!overlap) { HIPCHK(hipEventRecord(ev_ff, stream)); HIPCHK(hipStreamWaitEvent(tail, ev_ff, 0)); }
    } else {
        if (!d2f) {
            FirCcfParams f{};
            f.in = filt_in; f.out = r2f; f.q0 = n2_0; f.count = c2; f.taps = filt_taps.p; f.nt = filt_nt;
            f.port = side && out->filtered ? reinterpret_cast<float2*>(out->filtered) : nullptr;
            f.port_cap = side ? out->filtered_cap : 0;
            f.counts = counts;
            launch_fir_ccf(f, B, cs);
        }
        if (fam == F_QPSK || fam == F_BPSK) {
            // recursive chain + Viterbi below; nothing else at the sample rate
        } else if (fsk4_disc) {
            RingC r2l4{s2l.p, s2_mask}, r2g{s2g.p, s2_mask};
            Disc4fskParams d{}; d.in = r2f; d.out = r2l4; d.q0 = n2_0; d.count = c2; d.taps = disc4_taps.p; d.nt = disc4_nt;
            launch_disc_4fsk(d, B, cs);
            FirCcfParams f{}; f.in = r2l4; f.out = r2g; f.q0 = n2_0; f.count = c2; f.taps = sym4_taps.p; f.nt = sym4_nt;   // _symbol_filter
            launch_fir_ccf(f, B, cs);
        } else if (fam == F_GMSK || fam == F_4FSK || fm) {
            QuadDemodParams q{}; q.in = r2f; q.out = r2d; q.q0 = n2_0; q.count = c2; q.gain = demod_gain; q.atan_tab = atan_tab.p;
            launch_quad_demod(q, B, cs);
        } else {
            Disc2fskParams d{}; d.in = r2f; d.out = r2d; d.q0 = n2_0; d.count = c2; d.up = disc_up.p; d.lo = disc_lo.p; d.nt = disc_nt;
            launch_disc_2fsk(d, B, cs);
        }
        if (fam == F_QPSK || fam == F_BPSK || fsk4_disc) {
            // the tail reads r2f (written by k_fir_ccf above): it runs on the handle's own stream for these families
        } else {
            // r3 is what the previous call's tail (other stream) may still be reading
            if (!overlap && tail_pending) { HIPCHK(hipStreamWaitEvent(stream, ev_tail, 0)); tail_pending = false; }
            FirFffParams f{}; f.in = r2d; f.out = r3; f.q0 = n2_0; f.count = c2; f.taps = symf_taps.p; f.nt = symf_nt;
            launch_fir_fff(f, B, cs);
            if (!overlap) { HIPCHK(hipEventRecord(ev_ff, stream)); HIPCHK(hipStreamWaitEvent(tail, ev_ff, 0)); }
        }
    }
    // ---- stage D: symbol sync + FEC
    if (fam == F_QPSK || fam == F_BPSK || fsk4_disc) {
        QpskParams q{};
        q.in = fsk4_disc ? RingC{s2g.p, s2_mask} : r2f; q.np0 = n2_0; q.avail = n2_1; q.soft = RingB{soft.p, soft_mask}; q.st = qp_st.p;
        q.mmse = mmse_tab.p; q.tanh_tab = tanh_tab.p;
        q.c1_alpha = c1_alpha; q.c1_beta = c1_beta; q.c2_alpha = c2_alpha; q.c2_beta = c2_beta;
        q.ss_alpha = ss_alpha; q.ss_beta = ss_beta; q.ss_maxp = ss_maxp; q.ss_minp = ss_minp;
        q.rot = qp_rot; q.soft_mul = 48.0f; q.soft_add = 128.0f;
        if (fsk4_disc) { q.mode = 2; q.soft_mul = 128.0f; }   // gr_demod_4fsk.cpp:138-146,186-195
        if (fam == F_BPSK) {   // gr_demod_bpsk.cpp:54-62,67
            q.mode = 1; q.soft_mul = 64.0f;
            const float gain_omega = 0.005f;
            q.cr_gain_omega = gain_omega * gain_omega; q.cr_gain_mu = 0.05f;
            q.cr_omega_mid = (float)sps_eff; q.cr_omega_lim = 0.001f * (float)sps_eff;
        }
        q.port = side && out->constellation ? reinterpret_cast<float2*>(out->constellation) : nullptr;
        q.port_cap = side ? out->constellation_cap : 0;
        q.counts = counts;
        q.oo_snap = qp_snap.p + (size_t)slot * B;
        // recursion on `tail` behind this call's feed-forward kernels, decoder on `fecs` behind the recursion
        HIPCHK(hipEventRecord(ev_ff, stream));
        HIPCHK(hipStreamWaitEvent(tail, ev_ff, 0));
        if (q_valid[slot]) HIPCHK(hipStreamWaitEvent(tail, ev_fec[slot], 0));   // the soft ring holds two calls: decoder of call k - 2 done
        launch_qpsk_loops(q, B, tail);
        HIPCHK(hipEventRecord(ev_q[slot], tail));
        if (int rf = flush_fec(true)) return rf;                    // grouped order: the decoder of the call before goes with this recursion
        if (!grouped) HIPCHK(hipStreamWaitEvent(fecs, ev_q[slot], 0));
        FecParams f{};
        f.soft = RingB{soft.p, soft_mask};
        f.avail = q.oo_snap; f.avail_stride = sizeof(uint64_t); f.avail_mul = fam == F_BPSK ? 1 : 2;
        f.st = fec_st.p;
        f.bits_a = out ? out->bits_a : nullptr; f.bits_b = fam == F_BPSK && out ? out->bits_b : nullptr; f.bits_cap = out ? out->bits_cap : 0;
        f.counts = counts; f.branches = fam == F_BPSK ? 2 : 1;
        if (grouped) { fec_pending = f; fec_pending_slot = slot; fec_deferred = true; }
        else {
            launch_fec(f, B, fecs);
            HIPCHK(hipEventRecord(ev_fec[slot], fecs));
        }
        q_valid[slot] = true;
    } else {
        SymSyncParams s{};
        s.in = r3; s.avail = n2_1; s.soft = RingB{soft.p, soft_mask}; s.st = ss_st.p; s.mmse = mmse_tab.p;
        s.alpha = ss_alpha; s.beta = ss_beta; s.maxp = ss_maxp; s.minp = ss_minp;
        s.ted = fam == F_DMR && !m17 ? 0 : 1; s.soft_mul = 128.0f; s.soft_add = 128.0f;
        s.tail_scale = m17 ? 1.0f : 0.9f;
        // overlapped order: the 25 KB geometry (k_symsync_ff<16, 96>), whose workgroups fit beside two front-end workgroups on a CU.  The
        // 74 KB one was placed only as the front end of the next call drained -- 4.1 ms instead of 0.25, the decoder behind it, and the
        // front end after next waiting 0.5 ms per step for the ring this tail frees (profiles/r04_c1_timeline.log).
        s.slim = overlap ? 1 : 0;
        s.slicer = fam == F_DMR || fam == F_4FSK ? 1 : 0; s.tail = fam == F_DMR ? 1 : fam == F_4FSK ? 2 : 0;
        s.bits = out ? out->bits_a : nullptr; s.bits_cap = out ? out->bits_cap : 0;
        s.port = side && out->constellation ? reinterpret_cast<float2*>(out->constellation) : nullptr;
        s.port_cap = side ? out->constellation_cap : 0;
        s.counts = counts;
        if (!(QRL_DEV_SKIP & 4)) launch_symsync_ff(s, B, tail);
        if (fam == F_DMR && !m17 && dmo_out) {   // gr_dmr_dmo_sink on port 3 (= ring r3) of this call
            DmoParams dp{}; dp.in = r3; dp.q0 = n2_0; dp.count = (uint32_t)(n2_1 - n2_0); dp.st = dmo_st.p; dp.golay = dmo_golay.p;
            dp.out = dmo_out; dp.cap = dmo_cap; dp.counts = dmo_counts;
            launch_dmo_sink(dp, B, tail);
        }
        if (fam == F_DMR) { HIPCHK(hipEventRecord(ev_tail, tail)); tail_pending = true; }
        FecParams f{};
        f.soft = RingB{soft.p, soft_mask}; f.avail = &ss_st.p[0].oo; f.avail_stride = sizeof(SymSyncState); f.avail_mul = fam == F_4FSK ? 2 : 1; f.st = fec_st.p;
        f.bits_a = out ? out->bits_a : nullptr; f.bits_b = out ? out->bits_b : nullptr; f.bits_cap = out ? out->bits_cap : 0;
        f.counts = counts; f.branches = branches;
        if (fam != F_DMR) {
            if (!(QRL_DEV_SKIP & 8)) launch_fec(f, B, tail);
            HIPCHK(hipEventRecord(ev_tail, tail));
            tail_pending = true;
        }
        if (overlap) { HIPCHK(hipEventRecord(ev_tail2[slot], tail)); tail2_valid[slot] = true; }
    }
    HIPCHK(hipGetLastError());
    if (take_launch_error()) return QRL_ERR_HIP;   // (message already recorded by dyn_lds_limit)
    n_in = n_in1; n1 = n1_1; n2 = n2_1; ++call_no;
    return QRL_OK;
}

// everything of gr_demod_dsss behind the 1:50 stage (gr_demod_dsss.cpp:57-111); all on the handle's main stream
int qrl_demod::dsss_stages(uint64_t n2_0, uint64_t n2_1, const qrl_demod_out* out, uint32_t* counts, bool side)
{
    const int B = cfg.batch;
    RingC r2{s2.p, s2_mask}, ra{ds_ra.p, ds_mask}, rb{ds_rb.p, ds_mask}, rc{ds_rc.p, ds_mask}, rd{ds_rd.p, ds_mask}, rs{ds_sym.p, ds_sym_mask};
    const uint64_t n5_0 = n5, n5_1 = decim_count(n2_1, 13, 50);
    const uint32_t c5 = (uint32_t)(n5_1 - n5_0);
    {   // _resampler_if: rational_resampler_ccf(13, 50)
        ResampParams p{};
        p.in = nullptr; p.in_ring = r2; p.n0 = n2_0; p.n = (uint32_t)(n2_1 - n2_0);
        p.out = ra; p.q0 = n5_0; p.q_count = c5;
        p.taps = ds_rs.p; p.I = 13; p.D = 50; p.Jp = ds_Jp;
        launch_resamp(p, B, stream);
    }
    {   // _costas_freq
        DsssLoopParams p{}; p.in = ra; p.out = rb; p.q0 = n5_0; p.count = c5; p.st = ds_st.p; p.tanh_tab = tanh_tab.p; p.alpha = ds_a1; p.beta = ds_b1;
        launch_dsss_loop(p, 0, B, stream);
    }
    {   // _filter -> port 0
        FirCcfParams f{};
        f.in = rb; f.out = rc; f.q0 = n5_0; f.count = c5; f.taps = ds_filt.p; f.nt = ds_nf;
        f.port = side && out->filtered ? reinterpret_cast<float2*>(out->filtered) : nullptr;
        f.port_cap = side ? out->filtered_cap : 0;
        f.counts = counts;
        launch_fir_ccf(f, B, stream);
    }
    {   // _agc
        DsssLoopParams p{}; p.in = rc; p.out = rd; p.q0 = n5_0; p.count = c5; p.st = ds_st.p; p.tanh_tab = tanh_tab.p;
        launch_dsss_loop(p, 1, B, stream);
    }
    // _dsss_decoder: output I needs x[325 (I - 1) + 599]
    const uint64_t nsy_1 = n5_1 >= 275 ? (n5_1 - 275) / 325 + 1 : 0;
    {
        DsssMfParams p{}; p.in = rd; p.out = rs; p.i0 = nsy; p.count = (uint32_t)(nsy_1 - nsy); p.taps = ds_mf.p;
        launch_dsss_mf(p, B, stream);
    }
    {   // _clock_recovery -> _costas_loop (port 1) -> soft symbols
        DsssTailParams p{};
        p.in = rs; p.avail = nsy_1; p.soft = RingB{soft.p, soft_mask}; p.st = ds_tail.p; p.mmse = mmse_tab.p;
        const float gain_omega = 0.005f;
        p.gain_omega = gain_omega * gain_omega; p.gain_mu = 0.05f; p.omega_mid = 1.0f; p.omega_lim = 0.005f * 1.0f;
        p.alpha = ds_a2; p.beta = ds_b2;
        p.port = side && out->constellation ? reinterpret_cast<float2*>(out->constellation) : nullptr;
        p.port_cap = side ? out->constellation_cap : 0;
        p.counts = counts;
        launch_dsss_tail(p, B, stream);
    }
    FecParams f{};
    f.soft = RingB{soft.p, soft_mask};
    f.avail = &ds_tail.p[0].oo; f.avail_stride = sizeof(DsssTailState); f.avail_mul = 1;
    f.st = fec_st.p;
    f.bits_a = out ? out->bits_a : nullptr; f.bits_b = out ? out->bits_b : nullptr; f.bits_cap = out ? out->bits_cap : 0;
    f.counts = counts; f.branches = 2;
    launch_fec(f, B, stream);
    n5 = n5_1; nsy = nsy_1;
    return QRL_OK;
}

// everything of gr_demod_nbfm / gr_demod_am / gr_demod_wbfm behind the first resampler; all on the handle's main stream
int qrl_demod::analog_stages(uint64_t n2_0, uint64_t n2_1, const qrl_demod_out* out, uint32_t* counts, bool side)
{
    const int B = cfg.batch;
    RingC r2{s2.p, s2_mask}, r2f{s2f.p, s2_mask};
    const uint32_t c2 = (uint32_t)(n2_1 - n2_0);
    float2* fport = side && out->filtered ? reinterpret_cast<float2*>(out->filtered) : nullptr;
    const size_t fcap = side ? out->filtered_cap : 0;
    if (an_kind == 3) launch_scale_c(r2, n2_0, c2, an_if_gain, B, stream);   // _if_gain, gr_demod_ssb.cpp:45 (0.9 until gr_demod_ssb::set_gain)
    if (an_kind == 1 || an_kind == 3) {   // _filter -> port 0
        FirCccParams f{}; f.in = r2; f.out = r2f; f.q0 = n2_0; f.count = c2; f.taps = an_filt_c.p; f.nt = an_nfc;
        f.port = fport; f.port_cap = fcap; f.counts = counts;
        launch_an_fir_ccc(f, B, stream);
    } else {
        FirCcfParams f{}; f.in = r2; f.out = r2f; f.q0 = n2_0; f.count = c2; f.taps = filt_taps.p; f.nt = filt_nt;
        f.port = fport; f.port_cap = fcap; f.counts = counts;
        launch_fir_ccf(f, B, stream);
    }
    RingF f1{an_f1.p, an_m1}, f2{an_f2.p, an_m2}, f3{an_f3.p, an_m2};
    {   // _squelch and the recursions directly behind it
        AnGateParams g{}; g.in = r2f; g.out = f1; g.q0 = n2_0; g.count = c2; g.st = an_st.p; g.atan_tab = atan_tab.p;
        g.env = an_env.p; g.ramp = an_ramp; g.alpha = 0.01; g.one_minus_alpha = 1.0 - 0.01; g.threshold = an_threshold;
        g.gain = an_gain; g.attack = an_attack; g.decay = an_decay; g.ff0 = an_ff[0]; g.ff1 = an_ff[1]; g.fb1 = an_fb1;
        g.outc = RingC{an_c1.p, an_m1}; g.ref = 0.25f; g.clip = 0.95f;             // agc2_cc(0.1, 0.1, 0.25, 1), clipper_cc(0.95): gr_demod_ssb.cpp:52,58
        launch_an_gate(g, an_kind, B, stream);
    }
    float* aport = out ? out->audio : nullptr;
    const size_t acap = out ? out->audio_cap : 0;
    if (an_kind == 3) {   // _stretcher, _complex_to_real, _level_control, _audio_filter -> port 1 (whole chunks of 1024 gated items)
        const uint32_t max_chunked = c2 + 1024;
        AnStretchParams sp{}; sp.in = RingC{an_c1.p, an_m1}; sp.out = f2; sp.st = an_st.p; sp.level = 1.333f;
        launch_an_stretch(sp, max_chunked, B, stream);
        AnFirParams p{}; p.in = f2; p.out = RingF{nullptr, 0}; p.st = an_st.p; p.taps = an_ftaps.p; p.nt = an_nf; p.I = 0; p.D = 1;
        p.port = aport; p.port_cap = acap; p.counts = counts;
        launch_an_fir(p, max_chunked, B, stream);
        return QRL_OK;
    }
    const uint32_t max_out = (uint32_t)((uint64_t)c2 * an_I / an_D + 2);
    {   // _audio_resampler (WBFM: -> port 1)
        AnResampParams p{}; p.in = f1; p.out = f2; p.st = an_st.p; p.taps = an_rtaps.p; p.nt = an_nr; p.I = an_I; p.D = an_D;
        if (an_kind == 2) { p.port = aport; p.port_cap = acap; p.counts = counts; }
        launch_an_resamp(p, max_out, B, stream);
    }
    const bool ctcss = an_kind == 0 && ctcss_tone != 0.0f;
    if (ctcss) {   // _ctcss (gating): audio resampler -> ring f4, item count g2 (gr_demod_nbfm.cpp:110-119)
        CtcssParams p{}; p.in = f2; p.out = RingF{an_f4.p, an_m2}; p.st = an_st.p; p.cs = an_cs.p; p.I = an_I; p.D = an_D;
        for (int k = 0; k < 3; ++k) { p.wr[k] = ct_wr[k]; p.wi[k] = ct_wi[k]; }
        p.level = 0.01; p.len = 8000; p.ramp = 160; p.env = an_env_ct.p;
        launch_an_ctcss(p, B, stream);
    }
    if (an_kind != 2) {   // _audio_filter (AM: -> port 1)
        AnFirParams p{}; p.in = f2; p.out = f3; p.st = an_st.p; p.taps = an_ftaps.p; p.nt = an_nf; p.I = an_I; p.D = an_D;
        if (ctcss) { p.in = RingF{an_f4.p, an_m2}; p.taps = an_ftaps_ct.p; p.nt = an_nf_ct; p.cs = an_cs.p; }   // band_pass_2(1, 8000, 300, 3500, 200, 35, BH), :112-113
        if (an_kind == 1) { p.port = aport; p.port_cap = acap; p.counts = counts; }
        launch_an_fir(p, max_out, B, stream);
    }
    if (an_kind == 0) {   // _de_emph_filter, _level_control -> port 1
        AnDeemphParams p{}; p.in = f3; p.st = an_st.p; p.I = an_I; p.D = an_D; p.cs = ctcss ? an_cs.p : nullptr;
        p.ff0 = an_de_ff[0]; p.ff1 = an_de_ff[1]; p.fb1 = an_de_fb1; p.port = aport; p.port_cap = acap; p.counts = counts;
        launch_an_deemph(p, B, stream);
    }
    return QRL_OK;
}

// =============================================================================== C ABI
extern "C" {

const char* qrl_version(void) { return "qrl_hip 0.1 (gfx950)"; }
const char* qrl_last_error(void) { return g_last_error.c_str(); }
const char* qrl_strerror(int s)
{
    switch (s) {
    case QRL_OK: return "ok";
    case QRL_ERR_ARG: return "invalid argument or unsupported mode";
    case QRL_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU fallback)";
    case QRL_ERR_HIP: return "HIP runtime error";
    case QRL_ERR_NOMEM: return "out of device memory";
    case QRL_ERR_TOO_BIG: return "chunk larger than max_chunk";
    case QRL_ERR_STATE: return "invalid handle state";
    }
    return "unknown";
}

int qrl_init(int device, qrl_ctx** ctx)
{
    if (!ctx) return QRL_ERR_ARG;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(QRL_ERR_NO_DEVICE, "hipGetDeviceCount: no device");
    if (device < 0 || device >= count) return fail(QRL_ERR_NO_DEVICE, "device index out of range");
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0 && std::strncmp(prop.gcnArchName, "gfx94", 5) != 0)
        return fail(QRL_ERR_NO_DEVICE, std::string("unsupported architecture ") + prop.gcnArchName);
    *ctx = new (std::nothrow) qrl_ctx{device};
    return *ctx ? QRL_OK : QRL_ERR_NOMEM;
}
void qrl_shutdown(qrl_ctx* ctx) { delete ctx; }

int qrl_demod_create(qrl_ctx* ctx, const qrl_demod_config* cfg, qrl_demod** outp)
{
    if (!ctx || !cfg || !outp) return QRL_ERR_ARG;
    if (cfg->batch < 1 || cfg->max_chunk < 1) return fail(QRL_ERR_ARG, "batch and max_chunk must be >= 1");
    std::unique_ptr<qrl_demod> d(new (std::nothrow) qrl_demod);
    if (!d) return QRL_ERR_NOMEM;
    d->ctx = ctx;
    d->cfg = *cfg;
    qrl_demod_config& c = d->cfg;
    if (c.use_mode_defaults) {  // literals of gr_demod_base.cpp:203-210
        c.samp_rate = 1000000; c.carrier_freq = 1700;
        switch (c.modem_type) {
        case QRL_MODEM_2FSK2KFM:  c.sps = 5;  c.filter_width = 4000;  c.fm = 1; break;
        case QRL_MODEM_2FSK1KFM:  c.sps = 10; c.filter_width = 2500;  c.fm = 1; break;
        case QRL_MODEM_2FSK2K:    c.sps = 5;  c.filter_width = 4000;  c.fm = 0; break;
        case QRL_MODEM_2FSK1K:    c.sps = 10; c.filter_width = 2000;  c.fm = 0; break;
        case QRL_MODEM_2FSK10KFM: c.sps = 1;  c.filter_width = 25000; c.fm = 1; break;
        case QRL_MODEM_GMSK2K:    c.sps = 5;  c.filter_width = 4000;  c.fm = 0; break;
        case QRL_MODEM_GMSK1K:    c.sps = 10; c.filter_width = 2000;  c.fm = 0; break;
        case QRL_MODEM_GMSK10K:   c.sps = 1;  c.filter_width = 20000; c.fm = 0; break;
        case QRL_MODEM_QPSK250K:  c.sps = 2;  c.filter_width = 160000; c.fm = 0; break;   // gr_demod_base.cpp:223
        case QRL_MODEM_QPSKVIDEO: c.sps = 2;  c.filter_width = 160000; c.fm = 0; break;   // :224
        case QRL_MODEM_QPSK2K:    c.sps = 125; c.filter_width = 1300;  c.fm = 0; break;   // :221
        case QRL_MODEM_QPSK20K:   c.sps = 25;  c.filter_width = 6500;  c.fm = 0; break;   // :222
        case QRL_MODEM_4FSK2K:    c.sps = 5;  c.filter_width = 4000;   c.fm = 0; break;   // gr_demod_base.cpp:211
        case QRL_MODEM_4FSK2KFM:  c.sps = 5;  c.filter_width = 3000;   c.fm = 1; break;   // gr_demod_base.cpp:212
        case QRL_MODEM_4FSK1KFM:  c.sps = 10; c.filter_width = 2000;   c.fm = 1; break;   // :213
        case QRL_MODEM_4FSK10KFM: c.sps = 1;  c.filter_width = 20000;  c.fm = 1; break;   // :214
        case QRL_MODEM_4FSK100K:  c.sps = 2;  c.filter_width = 125000; c.fm = 1; break;   // :225
        case QRL_MODEM_BPSK1K:    c.sps = 10; c.filter_width = 1300;   c.fm = 0; break;   // :216
        case QRL_MODEM_BPSK2K:    c.sps = 5;  c.filter_width = 2400;   c.fm = 0; break;   // :217
        case QRL_MODEM_DMR:       c.sps = 5;  c.filter_width = 5000;   c.fm = 0; break;   // make_gr_demod_dmr(5, 1000000) gr_demod_base.cpp:253
        case QRL_MODEM_BPSK8:     c.sps = 25;  c.filter_width = 150;   c.fm = 0; break;
        case QRL_MODEM_NBFM2500:  c.sps = 125; c.filter_width = 2500;  c.fm = 0; break;   // make_gr_demod_nbfm(125, ., 1700, 2500) gr_demod_base.cpp:219
        case QRL_MODEM_NBFM5000:  c.sps = 125; c.filter_width = 5000;  c.fm = 0; break;   // :220
        case QRL_MODEM_WBFM:      c.sps = 125; c.filter_width = 75000; c.fm = 0; break;   // make_gr_demod_wbfm(125, ., 1700, 75000) :228
        case QRL_MODEM_AM5000:    c.sps = 125; c.filter_width = 5000;  c.fm = 0; break;
        case QRL_MODEM_USB2500: case QRL_MODEM_LSB2500: c.sps = 125; c.filter_width = 2700; c.fm = 0; break;   // make_gr_demod_ssb(125, ., 1700, 2700, sb) :226-227   // make_gr_demod_am(125, ., 1700, 5000) :215   // make_gr_demod_dsss(25, ., 1700, 150) gr_demod_base.cpp:218
        case QRL_MODEM_M17:       c.sps = 125; c.filter_width = 9000;  c.fm = 0; break;   // make_gr_demod_m17() gr_demod_base.cpp:252, defaults gr_demod_m17.h:41-42
        default: return fail(QRL_ERR_ARG, "modem_type not supported by this build");
        }
    }
    switch (c.modem_type) {
    case QRL_MODEM_2FSK2KFM: case QRL_MODEM_2FSK1KFM: case QRL_MODEM_2FSK2K: case QRL_MODEM_2FSK1K: case QRL_MODEM_2FSK10KFM:
        d->fam = qrl_demod::F_2FSK; break;
    case QRL_MODEM_GMSK2K: case QRL_MODEM_GMSK1K: case QRL_MODEM_GMSK10K:
        d->fam = qrl_demod::F_GMSK; break;
    case QRL_MODEM_QPSK250K: case QRL_MODEM_QPSKVIDEO: case QRL_MODEM_QPSK2K: case QRL_MODEM_QPSK20K:
        d->fam = qrl_demod::F_QPSK; break;
    case QRL_MODEM_DMR:
        d->fam = qrl_demod::F_DMR; break;
    case QRL_MODEM_M17:
        d->fam = qrl_demod::F_DMR; d->m17 = true; break;
    case QRL_MODEM_4FSK2K: case QRL_MODEM_4FSK2KFM: case QRL_MODEM_4FSK1KFM: case QRL_MODEM_4FSK10KFM: case QRL_MODEM_4FSK100K:
        d->fam = qrl_demod::F_4FSK; break;
    case QRL_MODEM_BPSK1K: case QRL_MODEM_BPSK2K:
        d->fam = qrl_demod::F_BPSK; break;
    case QRL_MODEM_BPSK8:
        d->fam = qrl_demod::F_DSSS; break;
    case QRL_MODEM_NBFM2500: case QRL_MODEM_NBFM5000:
        d->fam = qrl_demod::F_ANALOG; d->an_kind = 0; break;
    case QRL_MODEM_AM5000:
        d->fam = qrl_demod::F_ANALOG; d->an_kind = 1; break;
    case QRL_MODEM_WBFM:
        d->fam = qrl_demod::F_ANALOG; d->an_kind = 2; break;
    case QRL_MODEM_USB2500: case QRL_MODEM_LSB2500:
        d->fam = qrl_demod::F_ANALOG; d->an_kind = 3; d->an_lsb = c.modem_type == QRL_MODEM_LSB2500; break;
    default: return fail(QRL_ERR_ARG, "modem_type not supported by this build");
    }
    if (c.samp_rate != 1000000) return fail(QRL_ERR_ARG, "internal samp_rate must be 1000000 (gr_demod_base.cpp:21)");
    if (c.device_samp_rate != 1000000 && (c.device_samp_rate < 2000000 || c.device_samp_rate % 1000000))
        return fail(QRL_ERR_ARG, "device_samp_rate must be 1e6 or a multiple of 1e6 >= 2e6");
    HIPCHK(hipSetDevice(ctx->device));
    // Streams.  The tail stream has the highest priority and its kernels are small enough to take over the slot of ONE
    // retiring front-end workgroup.  (Reserving CUs for it with a CU mask was measured: it costs the front end ~18 %.)
    {
        if (c.hip_stream) d->stream = static_cast<hipStream_t>(c.hip_stream);
        else {
            int r0;
            if (std::getenv("QRL_CU_MAIN")) { if ((r0 = create_role_stream(&d->stream, 0, "MAIN"))) return r0; }
            else if (std::getenv("QRL_MAIN_PRIO_LOW")) {   // experiment: a CU-masked stream has no priority argument (it is a NORMAL stream), so the masked tail only outranks the
                int plo = 0, phi = 0;                      // front end when the front end's stream is created BELOW normal
                (void)hipDeviceGetStreamPriorityRange(&plo, &phi);
                HIPCHK(hipStreamCreateWithPriority(&d->stream, hipStreamNonBlocking, plo));
            }
            else HIPCHK(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
            d->own_stream = true;
        }
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        // THREE DIFFERENT PRIORITIES, and not for the scheduling: the runtime multiplexes the streams of one priority onto a few hardware
        // queues (least-used first), and two streams of a handle that land on the same queue run their kernels one after the other --
        // the overlapped and grouped orders then silently degrade to the serial one (seen as C4 4.5 instead of 3.05 ms and C2 2.2 instead
        // of 1.78 ms per step in the sub-lines of a long bench process, depending on how many streams the process had created and
        // destroyed before: tools/experiments/r04_subline_order*.py).  Queues of different priorities are never shared.
        int r1;
        if ((r1 = create_role_stream(&d->tail, hi, "TAIL")) || (r1 = create_role_stream(&d->fecs, lo, "FEC"))) return r1;
        // the front end's helper stream: the handle's own stream only (a caller's stream may share its hardware queue with anything), normal priority
        if (d->own_stream && !std::getenv("QRL_NO_PRE")) { if ((r1 = create_role_stream(&d->pre, 0, "PRE"))) return r1; }
    }
    HIPCHK(hipEventCreateWithFlags(&d->ev_ff, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&d->ev_pre, hipEventDisableTiming));
    for (auto& e : d->ev_fe) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&d->ev_tail, hipEventDisableTiming));
    for (auto& e : d->ev_tail2) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : d->ev_q) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : d->ev_fec) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    int r = d->build();
    if (r) return r;
    *outp = d.release();
    return QRL_OK;
}
void qrl_demod_destroy(qrl_demod* d) { if (d) { (void)d->sync_all(); delete d; } }

int qrl_demod_reset(qrl_demod* d)
{
    if (!d) return QRL_ERR_ARG;
    if (int rs = d->sync_all()) return rs;
    return d->init_state();
}
int qrl_demod_set_carrier_offset(qrl_demod* d, double hz)
{
    if (!d) return QRL_ERR_ARG;
    if (int rs = d->sync_all()) return rs;
    d->rot_acc += (d->n_in - d->rot_nbase) * d->rot_inc;  // phase-continuous
    d->rot_nbase = d->n_in;
    d->cfg.carrier_offset_hz = hz;
    d->rot_inc = phase_inc_to_turn(2 * M_PI * -hz / d->cfg.device_samp_rate);
    return d->upload_rot_table();
}
int qrl_demod_stream_wait(qrl_demod* d, void* hip_stream)
{
    if (!d) return QRL_ERR_ARG;
    if (int rf = d->flush_fec(false)) return rf;
    hipStream_t user = static_cast<hipStream_t>(hip_stream);
    if (!d->ev_user[0]) for (auto& e : d->ev_user) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIPCHK(hipEventRecord(d->ev_user[0], d->stream));
    HIPCHK(hipEventRecord(d->ev_user[1], d->tail));
    HIPCHK(hipEventRecord(d->ev_user[2], d->fecs));
    HIPCHK(hipEventRecord(d->ev_user[3], d->pre ? d->pre : d->stream));   // k_hist reads the caller's IQ on the helper stream
    for (auto e : d->ev_user) HIPCHK(hipStreamWaitEvent(user, e, 0));
    return QRL_OK;
}
int qrl_demod_set_dmo_output(qrl_demod* d, uint8_t* frames, size_t cap_frames, uint32_t* counts)
{
    if (!d) return QRL_ERR_ARG;
    if (d->fam != qrl_demod::F_DMR || d->m17) return qrl_set_error(QRL_ERR_ARG, "the DMO slicer sits behind port 3 of gr_demod_dmr: QRL_MODEM_DMR only");
    // kernel parameters are captured at launch: swapping the output pointers between calls needs no synchronisation (a host layer
    // that double-buffers its mailboxes calls this before every process).  Only the first use (state allocation) and switching
    // the block off wait for the work in flight.
    if (!frames) { if (int rs = d->sync_all()) return rs; d->dmo_out = nullptr; return QRL_OK; }
    if (!counts || cap_frames < 1 || cap_frames > 0xFFFFFFFFu) return QRL_ERR_ARG;
    int r;
    if (!d->dmo_st.p) {
        if (int rs = d->sync_all()) return rs;
        if ((r = d->dmo_st.alloc(d->cfg.batch)) || (r = d->dmo_golay.upload(golay1987_table()))) return r;
        std::vector<DmoState> ds(d->cfg.batch);
        for (auto& x : ds) { std::memset(&x, 0, sizeof x); x.endPtr = 9999; }
        if (hipMemcpy(d->dmo_st.p, ds.data(), ds.size() * sizeof(DmoState), hipMemcpyHostToDevice) != hipSuccess) return QRL_ERR_HIP;
    }
    d->dmo_out = frames; d->dmo_cap = (uint32_t)cap_frames; d->dmo_counts = counts;
    return QRL_OK;
}
int qrl_demod_set_option(qrl_demod* d, int option, int value)
{
    if (!d) return QRL_ERR_ARG;
    switch (option) {
    case QRL_OPT_OVERLAP:
        if (value != 0 && !d->overlap_capable) return qrl_set_error(QRL_ERR_ARG, "overlapped mode exists for the 2FSK family only");
        if (int rs = d->sync_all()) return rs;
        d->overlap = value != 0;
        d->tail2_valid[0] = d->tail2_valid[1] = false;
        return QRL_OK;
    case QRL_OPT_FLL_SLIM:
        if (int rs = d->sync_all()) return rs;
        d->fll_slim = value != 0;
        return QRL_OK;
    case QRL_OPT_GROUPED:
        if (value != 0 && !d->grouped_capable) return qrl_set_error(QRL_ERR_ARG, "the grouped order exists for the gr_demod_qpsk chain only");
        if (int rs = d->sync_all()) return rs;
        d->grouped = value != 0;
        return QRL_OK;
    case QRL_OPT_INPUT_RESIDENT: {
        if (int rs = d->sync_all()) return rs;
        if (value != 0 && d->pre) {   // the second edge scratch of the stage that reads the caller's IQ (the helper stages a call ahead)
            DecimStage& st = d->fe.used ? d->fe : d->first;
            if (st.edge_len && !st.edge_b.p && st.edge_b.alloc((size_t)d->cfg.batch * st.edge_len)) return qrl_set_error(QRL_ERR_HIP, "edge scratch");
        }
        d->input_resident = value != 0;
        return QRL_OK;
    }
    case QRL_OPT_UNFUSED_DEC2:
        if (!d->d2f_capable) return qrl_set_error(QRL_ERR_ARG, "this chain has no fused 1:2 decimator + shaping filter");
        if (d->n_in != 0) return qrl_set_error(QRL_ERR_STATE, "QRL_OPT_UNFUSED_DEC2 can only be set before the first sample (the two forms carry different state)");
        d->d2f = value == 0;
        return QRL_OK;
    default:
        return qrl_set_error(QRL_ERR_ARG, "unknown option");
    }
}
int qrl_demod_out_caps(const qrl_demod* d, size_t n, size_t* fcap, size_t* ccap, size_t* bcap)
{
    if (!d) return QRL_ERR_ARG;
    const size_t n1 = d->fe.used ? n / d->fe_decim + 2 : n;
    const size_t n2 = n1 * d->interp / d->decim + 2;
    const size_t ns = n2 / (size_t)(d->sps_eff > 1 ? d->sps_eff - 1 : 1) + 8;
    if (fcap) *fcap = n2;
    if (ccap) *ccap = ns;
    if (bcap) *bcap = d->fam == qrl_demod::F_DMR ? 2 * ns + 8 : d->fam == qrl_demod::F_QPSK || d->fam == qrl_demod::F_4FSK ? (ns / 80 + 2) * 80 : (ns / 2 / 80 + 2) * 80;
    return QRL_OK;
}
int qrl_demod_audio_cap(const qrl_demod* d, size_t n, size_t* audio_cap)
{
    if (!d || !audio_cap) return QRL_ERR_ARG;
    if (d->fam != qrl_demod::F_ANALOG) { *audio_cap = 0; return QRL_OK; }
    const size_t n1 = d->fe.used ? n / d->fe_decim + 2 : n;
    const size_t n2 = n1 * d->interp / d->decim + 2;
    *audio_cap = d->an_kind == 3 ? n2 + 1024 + 4 : n2 * d->an_I / d->an_D + 4;
    return QRL_OK;
}
int qrl_demod_set_squelch(qrl_demod* d, double db)
{
    if (!d || d->fam != qrl_demod::F_ANALOG) return QRL_ERR_ARG;
    d->an_threshold = std::pow(10.0, db / 10);   // pwr_squelch_cc::set_threshold
    return QRL_OK;
}
int qrl_demod_time_domain_cap(const qrl_demod* d, size_t n, size_t* cap)
{
    if (!d || !cap) return QRL_ERR_ARG;
    const size_t n1 = d->fe.used ? n / d->fe_decim + 2 : n;
    *cap = n1 / (size_t)d->scope_D + 2;
    return QRL_OK;
}
int qrl_demod_set_time_domain_output(qrl_demod* d, float* samples, size_t cap, uint32_t* counts)
{
    if (!d) return QRL_ERR_ARG;
    if (samples && !counts) return qrl_set_error(QRL_ERR_ARG, "qrl_demod_set_time_domain_output: counts [batch] required");
    HIPCHK(hipSetDevice(d->ctx->device));
    if (samples && !d->s_scope.p) {   // first use: the ring of the 100 ksps scope signal (one call + the stages' block granularity)
        if (int rs = d->sync_all()) return rs;
        const size_t max1 = d->fe.used ? d->cfg.max_chunk / d->fe_decim + 2 : d->cfg.max_chunk;
        d->scope_mask = pow2_at_least(max1 / (size_t)d->scope_D + 256) - 1;
        if (int r = d->s_scope.alloc((size_t)d->cfg.batch * (d->scope_mask + 1))) return qrl_set_error(r, "scope ring");
        // the tap starts with the samples of the next call: outputs are indexed from the stream's 1 Msps position
        d->n_scope = decim_count(d->fe.used ? d->n1 : d->n_in, 1, d->scope_D);
    }
    if (samples && !d->scope_out) d->n_scope = decim_count(d->fe.used ? d->n1 : d->n_in, 1, d->scope_D);   // (re-)enabled: skip what was not tapped
    d->scope_out = reinterpret_cast<float2*>(samples); d->scope_cap = cap; d->scope_counts = counts;
    return QRL_OK;
}
int qrl_demod_set_ctcss(qrl_demod* d, float tone_hz)
{
    if (!d || d->fam != qrl_demod::F_ANALOG || d->an_kind != 0) return qrl_set_error(QRL_ERR_ARG, "qrl_demod_set_ctcss: NBFM receivers only (gr_demod_nbfm::set_ctcss)");
    if (tone_hz < 0.0f || tone_hz > 1000.0f) return QRL_ERR_ARG;
    HIPCHK(hipSetDevice(d->ctx->device));
    const bool was_on = d->ctcss_tone != 0.0f, on = tone_hz != 0.0f;
    if (int rs = d->sync_all()) return rs;
    if (on && !d->an_cs.p) {   // first use: state, the gated ring, the band-pass audio filter, the envelope of ramp 160 (double: float item x double envelope)
        int r;
        const std::vector<float> ft = band_pass_2(1, 8000, 300, 3500, 200, 35, WIN_BLACKMAN_HARRIS);       // gr_demod_nbfm.cpp:112-113
        d->an_nf_ct = (int)ft.size();
        std::vector<double> env(161);
        for (int k = 0; k <= 160; ++k) env[k] = 0.5 - std::cos(M_PI * (double)k / 160.0) / 2.0;
        if ((r = d->an_ftaps_ct.upload(ft)) || (r = d->an_env_ct.upload(env)) || (r = d->an_cs.alloc(d->cfg.batch)) ||
            (r = d->an_f4.alloc((size_t)d->cfg.batch * (d->an_m2 + 1)))) return qrl_set_error(r, "ctcss buffers");
    }
    if (on) {   // ctcss_squelch_ff::set_frequency -> update_fft_params: the tone and its neighbours in the CTCSS table (2 % at the ends / off the table)
        static const float tones[38] = {67.0f, 71.9f, 74.4f, 77.0f, 79.7f, 82.5f, 85.4f, 88.5f, 91.5f, 94.8f, 97.4f, 100.0f, 103.5f, 107.2f, 110.9f, 114.8f,
                                        118.8f, 123.0f, 127.3f, 131.8f, 136.5f, 141.3f, 146.2f, 151.4f, 156.7f, 162.2f, 167.9f, 173.8f, 179.9f, 186.2f, 192.8f,
                                        203.5f, 210.7f, 218.1f, 225.7f, 233.6f, 241.8f, 250.3f};
        int i = -1;
        for (int k = 0; k < 38; ++k) if (tones[k] == tone_hz) i = k;
        const float f[3] = {(i == -1 || i == 0) ? (float)((double)tone_hz * 0.98) : tones[i - 1], tone_hz, (i == -1 || i == 37) ? (float)((double)tone_hz * 1.02) : tones[i + 1]};   // double literals, then narrowed (ctcss_squelch_ff_impl.cc compute_freqs)
        for (int k = 0; k < 3; ++k) {
            const float w = (float)(2.0 * M_PI * f[k] / 8000);
            d->ct_wr[k] = (float)(2.0 * (double)cosf(w));
            d->ct_wi[k] = sinf(w);
        }
    }
    d->ctcss_tone = tone_hz;
    // switching the block in or out re-wires the audio path (the reference disconnects / connects under lock()): the chain restarts
    // from a fresh state, like qrl_demod_reset; a new tone while it is on re-initialises the Goertzel filters only (set_frequency)
    if (was_on != on) return d->init_state();
    if (on) {
        std::vector<CtcssState> cs(d->cfg.batch);
        if (hipMemcpy(cs.data(), d->an_cs.p, cs.size() * sizeof(CtcssState), hipMemcpyDeviceToHost) != hipSuccess) return QRL_ERR_HIP;
        for (auto& x : cs) { for (int k = 0; k < 3; ++k) x.d1[k] = x.d2[k] = 0.0f; x.processed = 0; }
        if (hipMemcpy(d->an_cs.p, cs.data(), cs.size() * sizeof(CtcssState), hipMemcpyHostToDevice) != hipSuccess) return QRL_ERR_HIP;
    }
    return QRL_OK;
}
int qrl_demod_set_filter_width(qrl_demod* d, int width)
{
    if (!d || d->fam != qrl_demod::F_ANALOG)
        return qrl_set_error(QRL_ERR_ARG, "qrl_demod_set_filter_width: analogue receivers only (gr_demod_base::set_filter_width forwards to WBFM, AM, NBFM, USB, LSB)");
    // firdes' sanity checks: 0 < cutoff <= fs / 2; the SSB band starts at 200 Hz
    if (width <= 0 || 2 * width > d->target || (d->an_kind == 3 && width <= 200)) return qrl_set_error(QRL_ERR_ARG, "qrl_demod_set_filter_width: width out of range");
    HIPCHK(hipSetDevice(d->ctx->device));
    if (int rs = d->sync_all()) return rs;
    const double fs = d->target, w = width;
    int r = QRL_OK;
    // the setters do not repeat the constructors' designs (transition widths, design functions, the SSB audio filter's gain of 2)
    if (d->an_kind == 0 || d->an_kind == 2) {   // gr_demod_nbfm.cpp:82-90, gr_demod_wbfm.cpp:76-84
        const std::vector<float> f = low_pass(1, fs, w, 1200, WIN_BLACKMAN_HARRIS);
        d->filt_nt = (int)f.size();
        r = d->filt_taps.upload(f);
        d->an_gain = (float)(d->target / ((d->an_kind == 0 ? 4 : 2) * M_PI * width));
    } else if (d->an_kind == 1) {               // gr_demod_am.cpp:84-91
        const auto fc = complex_band_pass(1, fs, -w, w, 1200, WIN_BLACKMAN_HARRIS);
        d->an_nfc = (int)fc.size();
        r = d->an_filt_c.upload(to_f2(fc));
    } else {                                    // gr_demod_ssb.cpp:89-101
        const auto fc = d->an_lsb ? complex_band_pass_2(1, fs, -w, -200, 200, 90, WIN_BLACKMAN_HARRIS) : complex_band_pass_2(1, fs, 200, w, 200, 90, WIN_BLACKMAN_HARRIS);
        const std::vector<float> ft = band_pass_2(2, fs, 200, w, 200, 90, WIN_BLACKMAN_HARRIS);
        d->an_nfc = (int)fc.size(); d->an_nf = (int)ft.size();
        if (!(r = d->an_filt_c.upload(to_f2(fc)))) r = d->an_ftaps.upload(ft);
    }
    if (r) return qrl_set_error(r, "qrl_demod_set_filter_width: filter tables");
    d->cfg.filter_width = width;
    // The reference swaps the taps of a running graph under lock() / unlock(), at a sample position its scheduler decides; here the chain restarts
    // from a fresh state (like qrl_demod_reset; squelch, CTCSS and AGC settings are kept) -- what the tests compare is the chain built with the setter's designs.
    return d->init_state();
}
int qrl_demod_set_gain(qrl_demod* d, float value)
{
    if (!d || d->fam != qrl_demod::F_ANALOG || d->an_kind != 3) return qrl_set_error(QRL_ERR_ARG, "qrl_demod_set_gain: SSB receivers only (gr_demod_base::set_gain)");
    d->an_if_gain = value;   // multiply_const_cc::set_k: from the next call on, nothing restarts
    return QRL_OK;
}
int qrl_demod_set_agc(qrl_demod* d, float attack, float decay)
{
    if (!d || d->fam != qrl_demod::F_ANALOG || (d->an_kind != 1 && d->an_kind != 3)) return QRL_ERR_ARG;
    d->an_attack = attack; d->an_decay = decay;
    return QRL_OK;
}
int qrl_demod_process(qrl_demod* d, const float* iq, size_t stride, size_t n, const qrl_demod_out* out)
{
    if (!d || (!iq && n)) return QRL_ERR_ARG;
    HIPCHK(hipSetDevice(d->ctx->device));
    (void)take_launch_error();   // a mark left on this thread by an earlier call that returned before reading it must not fail this one
    return d->process(iq, stride, n, out);
}
int qrl_demod_sync(qrl_demod* d)
{
    if (!d) return QRL_ERR_ARG;
    if (int rs = d->sync_all()) return rs;
    return QRL_OK;
}
void* qrl_demod_stream(qrl_demod* d) { return d ? d->stream : nullptr; }
int qrl_demod_internal_streams(qrl_demod* d, void* out[3])
{
    if (!d || !out) return QRL_ERR_ARG;
    out[0] = d->stream; out[1] = d->tail; out[2] = d->fecs;
    return QRL_OK;
}

int qrl_demod_profile(qrl_demod* d, int enable)
{
    if (!d) return QRL_ERR_ARG;
    d->profiling = enable != 0;
    return QRL_OK;
}
int qrl_demod_profile_read(qrl_demod* d, double* kernel_ms, uint64_t* launches, const char** kernel_name)
{
    if (!d) return QRL_ERR_ARG;
    if (int rs = d->sync_all()) return rs;
    double total = 0;
    for (auto& e : d->prof_events) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, e.first, e.second));
        total += ms;
        (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second);
    }
    if (kernel_ms) *kernel_ms = total;
    if (launches) *launches = d->prof_events.size();
    if (kernel_name) {
        const DecimStage& st = d->fe.used ? d->fe : d->first;
        *kernel_name = (!d->fe.used && d->d2f) ? "k_dec2_fir"
                     : (d->fe.used || d->interp == 1) ? (st.pm ? "k_decim_pm" : st.pl ? "k_decim_plx" : st.mfma ? "k_decim_mfma" : "k_decim") : "k_resamp";
    }
    d->prof_events.clear();
    return QRL_OK;
}

/* developer aid (not part of the drop-in surface): phase profile of k_decim_mfma under QRL_DBG=32 */
void qrl_debug_decim_prof(unsigned long long* out8) { decim_mfma_prof_read(out8); }
void qrl_debug_decim_prof_enable(int on) { decim_mfma_prof_enable(on); }

int qrl_demod_process_host(qrl_demod* d, const float* iq_host, size_t stride, size_t n, uint8_t* bits_a_host,
                           uint8_t* bits_b_host, size_t bits_cap, uint32_t* counts_host)
{
    if (!d || !iq_host || !counts_host) return QRL_ERR_ARG;
    const size_t B = (size_t)d->cfg.batch;
    const size_t st = (n + 1) & ~(size_t)1;
    DevBuf<float2> iq; DevBuf<uint8_t> ba, bb; DevBuf<uint32_t> cnt;
    int r;
    if ((r = iq.alloc(B * st)) || (r = ba.alloc(B * bits_cap)) || (r = bb.alloc(B * bits_cap)) || (r = cnt.alloc(B * 4))) return r;
    HIPCHK(hipMemcpy2D(iq.p, st * sizeof(float2), iq_host, stride * sizeof(float2), n * sizeof(float2), B, hipMemcpyHostToDevice));
    qrl_demod_out o{};
    o.bits_a = ba.p; o.bits_b = bb.p; o.bits_cap = bits_cap; o.counts = cnt.p;
    if ((r = d->process(reinterpret_cast<const float*>(iq.p), st, n, &o))) return r;
    if (int rs = d->sync_all()) return rs;
    if (bits_a_host) HIPCHK(hipMemcpy(bits_a_host, ba.p, B * bits_cap, hipMemcpyDeviceToHost));
    if (bits_b_host) HIPCHK(hipMemcpy(bits_b_host, bb.p, B * bits_cap, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(counts_host, cnt.p, B * 4 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return QRL_OK;
}

// ---- host-only design helpers
static int copy_out(const std::vector<float>& v, float* dst) { if (dst) std::memcpy(dst, v.data(), v.size() * sizeof(float)); return (int)v.size(); }
int qrl_firdes_low_pass(double g, double fs, double fc, double tw, int w, float* t)
{ return t ? copy_out(low_pass(g, fs, fc, tw, (Window)w), t) : compute_ntaps(fs, tw, (Window)w); }
int qrl_firdes_low_pass_2(double g, double fs, double fc, double tw, double a, int w, float* t)
{ return t ? copy_out(low_pass_2(g, fs, fc, tw, a, (Window)w), t) : compute_ntaps_windes(fs, tw, a); }
int qrl_firdes_complex_band_pass(double g, double fs, double lo, double hi, double tw, int w, float* t)
{
    if (!t) return compute_ntaps(fs, tw, (Window)w);
    const auto v = complex_band_pass(g, fs, lo, hi, tw, (Window)w);
    std::memcpy(t, v.data(), v.size() * sizeof(std::complex<float>));
    return (int)v.size();
}
int qrl_firdes_root_raised_cosine(double g, double fs, double sr, double a, int n, float* t)
{ return t ? copy_out(root_raised_cosine(g, fs, sr, a, n), t) : (n | 1); }
int qrl_table_mmse(float* t) { return copy_out(mmse_table(), t); }
int qrl_table_atan(float* t) { return copy_out(atan_table(), t); }
int qrl_table_tanh(float* t) { return copy_out(tanh_table(), t); }
uint64_t qrl_phase_inc_to_turn(double r) { return phase_inc_to_turn(r); }

}  // extern "C"
