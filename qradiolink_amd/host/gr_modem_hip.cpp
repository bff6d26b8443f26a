// gr_modem_hip.cpp — see gr_modem_hip.h.  Every function cites the reference lines it restates.
#include "gr_modem_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <stdexcept>

namespace qrl_host {

namespace {
void chk(int rc, const char* what)
{
    if (rc != QRL_OK) throw std::runtime_error(std::string(what) + ": " + qrl_strerror(rc) + " (" + qrl_last_error() + ")");
}
void hchk(hipError_t e, const char* what)
{
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
enum {   // gr_modem_types (src/modem_types.h:5-50); the modes without a QRL_MODEM_* constant
    ModemTypeBPSK8 = 25, ModemTypeM17 = 40
};
bool is_1k(int m)   // the modes with the 8-bit FrameTypeVoice1 sync and no reserved byte (gr_modem.cpp:1143-1149, 1213-1217)
{
    return m == QRL_MODEM_BPSK1K || m == QRL_MODEM_2FSK1KFM || m == QRL_MODEM_2FSK1K || m == QRL_MODEM_GMSK1K || m == QRL_MODEM_4FSK1KFM;
}
}  // namespace

int modem_rx_frame_length(int m, int* bit_buf_len)   // gr_modem.cpp:203-322
{
    int len = 0, bits = 0;
    switch (m) {
    case QRL_MODEM_BPSK2K: case ModemTypeBPSK8: case QRL_MODEM_QPSK2K: case QRL_MODEM_4FSK2K: case QRL_MODEM_4FSK2KFM:
    case QRL_MODEM_2FSK2KFM: case QRL_MODEM_2FSK2K: case QRL_MODEM_GMSK2K: bits = 8 * 8; len = 7; break;
    case QRL_MODEM_BPSK1K: case QRL_MODEM_2FSK1KFM: case QRL_MODEM_2FSK1K: case QRL_MODEM_4FSK1KFM: case QRL_MODEM_GMSK1K: bits = 4 * 8; len = 4; break;
    case QRL_MODEM_QPSK20K: case QRL_MODEM_4FSK10KFM: case QRL_MODEM_2FSK10KFM: case QRL_MODEM_GMSK10K: bits = 48 * 8; len = 47; break;
    case QRL_MODEM_QPSKVIDEO: bits = 3123 * 8; len = 3122; break;
    case QRL_MODEM_QPSK250K: bits = 1517 * 8; len = 1516; break;
    case QRL_MODEM_4FSK100K: bits = 623 * 8; len = 622; break;
    case ModemTypeM17: bits = 46 * 8; len = 46; break;
    case QRL_MODEM_DMR: bits = 46 * 8; len = 9; break;
    default: break;
    }
    if (bit_buf_len) *bit_buf_len = bits;
    return len;
}
int modem_tx_frame_length(int m)   // gr_modem.cpp:105-199
{
    if (m == ModemTypeM17) return 16;
    return modem_rx_frame_length(m, nullptr);
}
bool modem_two_branches(int m)   // gr_modem.cpp:1048-1058
{
    switch (m) {
    case QRL_MODEM_BPSK2K: case QRL_MODEM_2FSK2KFM: case QRL_MODEM_2FSK2K: case QRL_MODEM_2FSK10KFM: case QRL_MODEM_GMSK2K: case QRL_MODEM_GMSK1K:
    case QRL_MODEM_GMSK10K: case QRL_MODEM_BPSK1K: case ModemTypeBPSK8: case QRL_MODEM_2FSK1KFM: case QRL_MODEM_2FSK1K: return true;
    default: return false;
    }
}

// ================================================================================================ gr_demod_base_hip
struct gr_demod_base_hip::slot {
    gr_complex* h_iq = nullptr;                           // pinned [streams][chunk]
    float* d_iq = nullptr;                                // device [streams][chunk] cf32
    bool const_copied = true;
    float *d_filt = nullptr, *d_const = nullptr; uint8_t *d_a = nullptr, *d_b = nullptr, *d_dmo = nullptr; uint32_t *d_cnt = nullptr, *d_dmocnt = nullptr;
    gr_complex* h_const = nullptr; uint8_t *h_a = nullptr, *h_b = nullptr, *h_dmo = nullptr; uint32_t *h_cnt = nullptr, *h_dmocnt = nullptr;   // pinned
    float *d_rssi = nullptr, *h_rssi = nullptr; bool rssi_valid = false;   // latest rssi_block value per stream (device / pinned)
    float *d_audio = nullptr, *h_audio = nullptr;         // analogue modes: port 1 (device / pinned)
    float *d_scope = nullptr; gr_complex* h_scope = nullptr; uint32_t *d_scnt = nullptr, *h_scnt = nullptr; bool scoped = false;   // time-domain scope items of the call
    uint8_t *d_fr[2] = {nullptr, nullptr}, *h_fr[2] = {nullptr, nullptr}; uint32_t *d_frcnt[2] = {nullptr, nullptr}, *h_frcnt[2] = {nullptr, nullptr};   // framed records of bits A / B
    bool framed = false, bits_copied = true;
    hipEvent_t done = nullptr;
};
static constexpr size_t kDmoCap = 16;

gr_demod_base_hip::gr_demod_base_hip(qrl_runtime& rt, int streams, int device_samp_rate, double carrier_offset_hz, size_t max_chunk)
    : d_rt(rt), d_n(streams), d_rate(device_samp_rate), d_offset(carrier_offset_hz), d_chunk(max_chunk & ~(size_t)1),
      d_boxa(streams), d_box1(streams), d_box2(streams), d_boxc(streams), d_boxd(streams)
{
    d_boxf[0].resize(streams); d_boxf[1].resize(streams); d_boxs.resize(streams);
    for (int k = 0; k < 2; ++k) { d_fbits[k].assign(streams, 0); d_fact[k].assign(streams, 0); }
    if (streams < 1 || d_chunk < 2) throw std::invalid_argument("gr_demod_base_hip: streams >= 1, max_chunk >= 2");
    hipStream_t s;
    // lowest priority -- not for the scheduling: streams of one priority share a few hardware queues, and a wait queued on this
    // stream would otherwise hold back the kernels of a handle stream that happens to sit on the same queue (csrc/engine.cpp, stream creation)
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    hchk(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio_lo), "hipStreamCreate");
    d_copy = s;
}
gr_demod_base_hip::~gr_demod_base_hip()
{
    try { flush(); } catch (...) {}
    close();
    if (d_copy) (void)hipStreamDestroy(static_cast<hipStream_t>(d_copy));
}
void gr_demod_base_hip::close()
{
    if (d_h) { qrl_demod_destroy(d_h); d_h = nullptr; }
    if (d_rssi) { qrl_rssi_destroy(d_rssi); d_rssi = nullptr; }
    if (d_fft) { qrl_fft_destroy(d_fft); d_fft = nullptr; }
    if (d_fftout) { (void)hipFree(d_fftout); d_fftout = nullptr; }
    for (auto& f : d_fs) if (f) { qrl_framesync_destroy(f); f = nullptr; }
    for (auto& sp : d_slot) {
        if (!sp) continue;
        if (sp->d_scope) (void)hipFree(sp->d_scope);
        if (sp->d_scnt) (void)hipFree(sp->d_scnt);
        if (sp->h_scope) (void)hipHostFree(sp->h_scope);
        if (sp->h_scnt) (void)hipHostFree(sp->h_scnt);
        for (int k = 0; k < 2; ++k) {
            if (sp->d_fr[k]) (void)hipFree(sp->d_fr[k]);
            if (sp->d_frcnt[k]) (void)hipFree(sp->d_frcnt[k]);
            if (sp->h_fr[k]) (void)hipHostFree(sp->h_fr[k]);
            if (sp->h_frcnt[k]) (void)hipHostFree(sp->h_frcnt[k]);
        }
        if (sp->d_rssi) (void)hipFree(sp->d_rssi);
        if (sp->h_rssi) (void)hipHostFree(sp->h_rssi);
        if (sp->d_audio) (void)hipFree(sp->d_audio);
        if (sp->h_audio) (void)hipHostFree(sp->h_audio);
        for (void* p : {(void*)sp->d_iq, (void*)sp->d_filt, (void*)sp->d_const, (void*)sp->d_a, (void*)sp->d_b, (void*)sp->d_dmo, (void*)sp->d_cnt, (void*)sp->d_dmocnt})
            if (p) (void)hipFree(p);
        for (void* p : {(void*)sp->h_iq, (void*)sp->h_const, (void*)sp->h_a, (void*)sp->h_b, (void*)sp->h_dmo, (void*)sp->h_cnt, (void*)sp->h_dmocnt})
            if (p) (void)hipHostFree(p);
        if (sp->done) (void)hipEventDestroy(sp->done);
        delete sp;
        sp = nullptr;
    }
    d_inflight = -1;
}
void gr_demod_base_hip::open()
{
    close();
    qrl_demod_config c{};
    c.modem_type = d_mode; c.use_mode_defaults = 1; c.device_samp_rate = d_rate; c.carrier_offset_hz = d_offset;
    c.batch = d_n; c.max_chunk = d_chunk; c.enable_side_outputs = 1;
    c.time_domain_samp_rate = d_scope_rate; c.time_domain_filter_width = d_scope_fw;
    chk(qrl_demod_create(d_rt.ctx(), &c, &d_h), "qrl_demod_create");
    chk(qrl_demod_out_caps(d_h, d_chunk, &d_fcap, &d_ccap, &d_bcap), "qrl_demod_out_caps");
    chk(qrl_demod_audio_cap(d_h, d_chunk, &d_acap), "qrl_demod_audio_cap");
    if (d_acap) chk(qrl_demod_set_squelch(d_h, (double)d_squelch), "qrl_demod_set_squelch");
    if (d_acap && d_mode == QRL_MODEM_AM5000) chk(qrl_demod_set_agc(d_h, d_agc_attack, d_agc_decay), "qrl_demod_set_agc");
    if (d_acap && d_ctcss != 0.0f && (d_mode == QRL_MODEM_NBFM2500 || d_mode == QRL_MODEM_NBFM5000)) chk(qrl_demod_set_ctcss(d_h, d_ctcss), "qrl_demod_set_ctcss");   // the reference's instances keep their tone across mode changes
    if (d_acap && d_width.count(d_mode)) chk(qrl_demod_set_filter_width(d_h, d_width[d_mode]), "qrl_demod_set_filter_width");   // ... and their set_filter_width designs
    if (d_acap && d_if_gain >= 0.0f && (d_mode == QRL_MODEM_USB2500 || d_mode == QRL_MODEM_LSB2500)) chk(qrl_demod_set_gain(d_h, d_if_gain), "qrl_demod_set_gain");
    const size_t N = (size_t)d_n;
    // side outputs on the copy stream: rssi_block behind port 0, rx_fft_c on the device-rate IQ (gr_demod_base.cpp:166,185,199-200)
    chk(qrl_rssi_create(d_rt.ctx(), d_n, d_rssi_cal, d_copy, &d_rssi), "qrl_rssi_create");
    chk(qrl_fft_create(d_rt.ctx(), d_n, d_fftsize, 5 /* WIN_BLACKMAN_HARRIS */, d_copy, &d_fft), "qrl_fft_create");
    chk(qrl_fft_set_enabled(d_fft, d_fft_on ? 1 : 0), "qrl_fft_set_enabled");
    d_level.assign(N, 0.0f);
    // L1 frame synchronisers on the copy stream, behind the demodulator (digital modes with gr_modem framing: not DMR, not analogue)
    const bool fs_on = d_want_fs && !d_acap && d_mode != QRL_MODEM_DMR;
    if (fs_on) {
        for (auto& f : d_fs) chk(qrl_framesync_create(d_rt.ctx(), d_mode, d_n, d_copy, &f), "qrl_framesync_create");
        const size_t fb = (size_t)qrl_framesync_frame_bytes(d_fs[0]);
        d_frcap = (d_bcap / 8 + fb + 96 + 16 * (d_bcap / std::max<size_t>(8 * fb, 8) + 2) + 3) & ~(size_t)3;   // never overflows (include/qrl_hip.h)
    }
    chk(qrl_demod_time_domain_cap(d_h, d_chunk, &d_scap), "qrl_demod_time_domain_cap");
    for (auto& sp : d_slot) {
        sp = new slot;
        hchk(hipMalloc(reinterpret_cast<void**>(&sp->d_scope), N * d_scap * sizeof(gr_complex)), "hipMalloc");
        hchk(hipMalloc(reinterpret_cast<void**>(&sp->d_scnt), N * sizeof(uint32_t)), "hipMalloc");
        hchk(hipHostMalloc(reinterpret_cast<void**>(&sp->h_scope), N * d_scap * sizeof(gr_complex), hipHostMallocDefault), "hipHostMalloc");
        hchk(hipHostMalloc(reinterpret_cast<void**>(&sp->h_scnt), N * sizeof(uint32_t), hipHostMallocDefault), "hipHostMalloc");
        if (fs_on) for (int k = 0; k < 2; ++k) {
            hchk(hipMalloc(reinterpret_cast<void**>(&sp->d_fr[k]), N * d_frcap), "hipMalloc");
            hchk(hipMalloc(reinterpret_cast<void**>(&sp->d_frcnt[k]), N * 3 * sizeof(uint32_t)), "hipMalloc");   // [2 N]: bytes, frames; [N]: bits collected under a sync
            hchk(hipHostMalloc(reinterpret_cast<void**>(&sp->h_fr[k]), N * d_frcap, hipHostMallocDefault), "hipHostMalloc");
            hchk(hipHostMalloc(reinterpret_cast<void**>(&sp->h_frcnt[k]), N * 3 * sizeof(uint32_t), hipHostMallocDefault), "hipHostMalloc");
        }
        hchk(hipMalloc(reinterpret_cast<void**>(&sp->d_rssi), N * sizeof(float)), "hipMalloc");
        hchk(hipHostMalloc(reinterpret_cast<void**>(&sp->h_rssi), N * sizeof(float), hipHostMallocDefault), "hipHostMalloc");
        hchk(hipHostMalloc(reinterpret_cast<void**>(&sp->h_iq), N * d_chunk * sizeof(gr_complex), hipHostMallocDefault), "hipHostMalloc");
        hchk(hipMalloc(reinterpret_cast<void**>(&sp->d_iq), N * d_chunk * sizeof(gr_complex)), "hipMalloc");
        hchk(hipMalloc(reinterpret_cast<void**>(&sp->d_filt), N * d_fcap * sizeof(gr_complex)), "hipMalloc");
        hchk(hipMalloc(reinterpret_cast<void**>(&sp->d_const), N * d_ccap * sizeof(gr_complex)), "hipMalloc");
        hchk(hipMalloc(reinterpret_cast<void**>(&sp->d_a), N * d_bcap), "hipMalloc");
        hchk(hipMalloc(reinterpret_cast<void**>(&sp->d_b), N * d_bcap), "hipMalloc");
        hchk(hipMalloc(reinterpret_cast<void**>(&sp->d_cnt), N * 4 * sizeof(uint32_t)), "hipMalloc");
        hchk(hipHostMalloc(reinterpret_cast<void**>(&sp->h_const), N * d_ccap * sizeof(gr_complex), hipHostMallocDefault), "hipHostMalloc");
        hchk(hipHostMalloc(reinterpret_cast<void**>(&sp->h_a), N * d_bcap, hipHostMallocDefault), "hipHostMalloc");
        hchk(hipHostMalloc(reinterpret_cast<void**>(&sp->h_b), N * d_bcap, hipHostMallocDefault), "hipHostMalloc");
        hchk(hipHostMalloc(reinterpret_cast<void**>(&sp->h_cnt), N * 4 * sizeof(uint32_t), hipHostMallocDefault), "hipHostMalloc");
        if (d_acap) {
            hchk(hipMalloc(reinterpret_cast<void**>(&sp->d_audio), N * d_acap * sizeof(float)), "hipMalloc");
            hchk(hipHostMalloc(reinterpret_cast<void**>(&sp->h_audio), N * d_acap * sizeof(float), hipHostMallocDefault), "hipHostMalloc");
        }
        if (d_mode == QRL_MODEM_DMR) {
            hchk(hipMalloc(reinterpret_cast<void**>(&sp->d_dmo), N * kDmoCap * QRL_DMO_RECORD_BYTES), "hipMalloc");
            hchk(hipMalloc(reinterpret_cast<void**>(&sp->d_dmocnt), N * sizeof(uint32_t)), "hipMalloc");
            hchk(hipHostMalloc(reinterpret_cast<void**>(&sp->h_dmo), N * kDmoCap * QRL_DMO_RECORD_BYTES, hipHostMallocDefault), "hipHostMalloc");
            hchk(hipHostMalloc(reinterpret_cast<void**>(&sp->h_dmocnt), N * sizeof(uint32_t), hipHostMallocDefault), "hipHostMalloc");
        }
        hchk(hipEventCreateWithFlags(&sp->done, hipEventDisableTiming), "hipEventCreate");
    }
    d_calls = 0;
}
void gr_demod_base_hip::set_mode(int mode)   // gr_demod_base::set_mode (src/gr/gr_demod_base.cpp:460-1090): swap the graph, drop what was queued
{
    std::lock_guard<std::recursive_mutex> hg(d_hmutex);
    flush();
    d_mode = mode;
    try { open(); }
    catch (const std::exception& e) {
        if (d_scope_rate == 0 && d_scope_fw == 0.0) throw;
        d_scope_rate = 0; d_scope_fw = 0.0;   // set before a handle existed and only rejected now: back to the constructor's 1:10 tap
        open();
        throw std::invalid_argument(std::string("gr_demod_base_hip::set_mode: time-domain settings rejected and reset to the defaults (the mode is open): ") + e.what());
    }
    std::lock_guard<std::mutex> g(d_mutex);
    for (int s = 0; s < d_n; ++s) { d_box1[s].clear(); d_box2[s].clear(); d_boxc[s].clear(); d_boxd[s].clear(); d_boxa[s].clear(); d_boxf[0][s].clear(); d_boxf[1][s].clear(); d_boxs[s].clear(); d_fbits[0][s] = d_fbits[1][s] = d_fact[0][s] = d_fact[1][s] = 0; }
}
void gr_demod_base_hip::enable_device_framing(bool value)
{
    std::lock_guard<std::recursive_mutex> hg(d_hmutex);
    d_want_fs = value;
}
bool gr_demod_base_hip::takeFrames(int nr, int stream, std::vector<frame_record>& frames, size_t& bits, size_t& collected)
{
    std::lock_guard<std::mutex> g(d_mutex);
    const int k = nr == 2 ? 1 : 0;
    bits = d_fbits[k][stream]; collected = d_fact[k][stream];
    if (bits < 32) return false;   // gr_bit_sink::get_data (src/gr/gr_bit_sink.cpp:45-59): nothing below 32 bits, and nothing is consumed
    frames.clear();
    frames.swap(d_boxf[k][stream]);
    d_fbits[k][stream] = 0; d_fact[k][stream] = 0;
    return true;
}
size_t gr_demod_base_hip::peekFrameBits(int nr, int stream)
{
    std::lock_guard<std::mutex> g(d_mutex);
    return d_fbits[nr == 2 ? 1 : 0][stream];
}
std::vector<gr_demod_base_hip::frame_record> gr_demod_base_hip::getFrames(int nr, int stream)
{
    std::lock_guard<std::mutex> g(d_mutex);
    std::vector<frame_record> out;
    out.swap(d_boxf[nr == 2 ? 1 : 0][stream]);
    return out;
}
void gr_demod_base_hip::set_carrier_offset(double hz)   // gr_demod_base.cpp:1220-1225
{
    std::lock_guard<std::recursive_mutex> hg(d_hmutex);
    d_offset = hz;
    if (d_h) chk(qrl_demod_set_carrier_offset(d_h, hz), "qrl_demod_set_carrier_offset");
}
void gr_demod_base_hip::set_samp_rate(int device_samp_rate)   // gr_demod_base.cpp:1303-1362: the resampler is rebuilt
{
    std::lock_guard<std::recursive_mutex> hg(d_hmutex);
    flush();
    d_rate = device_samp_rate;
    if (d_mode >= 0) open();
}
void gr_demod_base_hip::work(const gr_complex* const* iq, size_t n)
{
    std::lock_guard<std::recursive_mutex> hg(d_hmutex);
    if (!d_h) throw std::runtime_error("gr_demod_base_hip::work before set_mode");
    if (n == 0) return;
    if (n > d_chunk || (n & 1)) throw std::invalid_argument("gr_demod_base_hip::work: n must be even and <= max_chunk");
    if (!d_demod_on && !d_fft_on) return;                        // _demod_valve closed and nobody else listens: the samples are dropped
    const int cur = (int)(d_calls & 1);
    slot& sl = *d_slot[cur];
    // slot `cur` was last used by call k - 2, which work(k - 1) has harvested: its buffers are free
    for (int s = 0; s < d_n; ++s) std::memcpy(sl.h_iq + (size_t)s * d_chunk, iq[s], n * sizeof(gr_complex));
    hipStream_t hs = static_cast<hipStream_t>(qrl_demod_stream(d_h)), cs = static_cast<hipStream_t>(d_copy);
    if (!d_demod_on) {                                           // _demod_valve closed (gr_demod_base.cpp:1150-1153): only the spectrum tap, which sits in front of it
        // everything of this call on the spectrum block's own stream (d_copy): the upload, the FFT fill behind it, and the host waits for THAT
        // stream before the slot is reused (the upload used to go on the demodulator's stream, which nothing ordered the FFT behind)
        hchk(hipMemcpyAsync(sl.d_iq, sl.h_iq, (size_t)d_n * d_chunk * sizeof(gr_complex), hipMemcpyHostToDevice, cs), "H2D");
        chk(qrl_fft_process(d_fft, sl.d_iq, d_chunk, n), "qrl_fft_process");
        hchk(hipStreamSynchronize(cs), "hipStreamSynchronize");  // (d_calls does not advance: the next call takes the same slot; no harvest belongs to this call)
        return;
    }
    hchk(hipMemcpyAsync(sl.d_iq, sl.h_iq, (size_t)d_n * d_chunk * sizeof(gr_complex), hipMemcpyHostToDevice, hs), "H2D");
    if (d_mode == QRL_MODEM_DMR) chk(qrl_demod_set_dmo_output(d_h, sl.d_dmo, kDmoCap, sl.d_dmocnt), "qrl_demod_set_dmo_output");
    sl.scoped = d_scope_on;
    chk(qrl_demod_set_time_domain_output(d_h, d_scope_on ? sl.d_scope : nullptr, d_scap, d_scope_on ? sl.d_scnt : nullptr), "qrl_demod_set_time_domain_output");
    qrl_demod_out o{};
    o.filtered = sl.d_filt; o.filtered_cap = d_fcap; o.constellation = sl.d_const; o.constellation_cap = d_ccap;
    o.bits_a = sl.d_a; o.bits_b = sl.d_b; o.bits_cap = d_bcap; o.counts = sl.d_cnt;
    o.audio = sl.d_audio; o.audio_cap = d_acap;
    chk(qrl_demod_process(d_h, sl.d_iq, d_chunk, n, &o), "qrl_demod_process");
    chk(qrl_demod_stream_wait(d_h, cs), "qrl_demod_stream_wait");
    const size_t N = (size_t)d_n;
    if (d_acap) hchk(hipMemcpyAsync(sl.h_audio, sl.d_audio, N * d_acap * sizeof(float), hipMemcpyDeviceToHost, cs), "D2H");
    hchk(hipMemcpyAsync(sl.h_cnt, sl.d_cnt, N * 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, cs), "D2H");
    if (sl.scoped) {
        hchk(hipMemcpyAsync(sl.h_scnt, sl.d_scnt, N * sizeof(uint32_t), hipMemcpyDeviceToHost, cs), "D2H");
        hchk(hipMemcpyAsync(sl.h_scope, sl.d_scope, N * d_scap * sizeof(gr_complex), hipMemcpyDeviceToHost, cs), "D2H");
    }
    sl.framed = d_fs[0] != nullptr;
    sl.bits_copied = !sl.framed || d_keep_bits;
    if (sl.framed) {
        // the frame synchronisers run on the copy stream behind the demodulator (no host synchronisation in between): bits A / B of
        // this call -> records { type, nbytes | _modem_sync << 16, payload }; only those travel to the host
        for (int k = 0; k < 2; ++k) chk(qrl_framesync_set_activity_output(d_fs[k], sl.d_frcnt[k] + 2 * N), "qrl_framesync_set_activity_output");
        chk(qrl_framesync_process(d_fs[0], sl.d_a, d_bcap, d_bcap, sl.d_cnt + 2, 4, sl.d_fr[0], d_frcap, sl.d_frcnt[0]), "qrl_framesync_process");
        chk(qrl_framesync_process(d_fs[1], sl.d_b, d_bcap, d_bcap, sl.d_cnt + 3, 4, sl.d_fr[1], d_frcap, sl.d_frcnt[1]), "qrl_framesync_process");
        for (int k = 0; k < 2; ++k) {
            hchk(hipMemcpyAsync(sl.h_frcnt[k], sl.d_frcnt[k], N * 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, cs), "D2H");
            hchk(hipMemcpyAsync(sl.h_fr[k], sl.d_fr[k], N * d_frcap, hipMemcpyDeviceToHost, cs), "D2H");
        }
    }
    if (sl.bits_copied) {
        hchk(hipMemcpyAsync(sl.h_a, sl.d_a, N * d_bcap, hipMemcpyDeviceToHost, cs), "D2H");
        hchk(hipMemcpyAsync(sl.h_b, sl.d_b, N * d_bcap, hipMemcpyDeviceToHost, cs), "D2H");
    }
    sl.const_copied = d_const_on;
    if (d_const_on) hchk(hipMemcpyAsync(sl.h_const, sl.d_const, N * d_ccap * sizeof(gr_complex), hipMemcpyDeviceToHost, cs), "D2H");
    if (d_mode == QRL_MODEM_DMR) {
        hchk(hipMemcpyAsync(sl.h_dmocnt, sl.d_dmocnt, N * sizeof(uint32_t), hipMemcpyDeviceToHost, cs), "D2H");
        hchk(hipMemcpyAsync(sl.h_dmo, sl.d_dmo, N * kDmoCap * QRL_DMO_RECORD_BYTES, hipMemcpyDeviceToHost, cs), "D2H");
    }
    sl.rssi_valid = d_rssi_on;
    if (d_rssi_on) {   // rssi_valve open: port 0 of this call through rssi_block, the probe keeps the latest value
        chk(qrl_rssi_process(d_rssi, sl.d_filt, d_fcap, d_fcap, sl.d_cnt, 4, nullptr, 0, sl.d_rssi, nullptr), "qrl_rssi_process");
        hchk(hipMemcpyAsync(sl.h_rssi, sl.d_rssi, N * sizeof(float), hipMemcpyDeviceToHost, cs), "D2H");
    }
    if (d_fft_on) chk(qrl_fft_process(d_fft, sl.d_iq, d_chunk, n), "qrl_fft_process");
    hchk(hipEventRecord(sl.done, cs), "hipEventRecord");
    const int prev = d_inflight;
    d_inflight = cur;
    ++d_calls;
    if (prev >= 0) harvest(prev);   // the GPU already has call k queued while the host waits for call k - 1
}
void gr_demod_base_hip::harvest(int which)
{
    slot& sl = *d_slot[which];
    hchk(hipEventSynchronize(sl.done), "hipEventSynchronize");
    std::lock_guard<std::mutex> g(d_mutex);
    for (int s = 0; s < d_n; ++s) {
        const uint32_t* c = sl.h_cnt + 4 * (size_t)s;
        if (sl.rssi_valid && c[0]) d_level[s] = sl.h_rssi[s];
        // gr_sample_sink::work (src/gr/gr_sample_sink.cpp:66-89): while more than 524288 items wait, the new ones are dropped
        if (sl.scoped && d_boxs[s].size() <= 524288) {
            const gr_complex* p = sl.h_scope + (size_t)s * d_scap;
            d_boxs[s].insert(d_boxs[s].end(), p, p + std::min<size_t>(sl.h_scnt[s], d_scap));
        }
        if (d_acap) {   // analogue modes: port 1 is audio (gr_audio_sink), no bit / constellation ports
            // gr_audio_sink::work (src/gr/gr_audio_sink.cpp:68-90): more than one second waiting = the reader is too slow: drop it all
            if (d_boxa[s].size() > 8000) d_boxa[s].clear();
            else d_boxa[s].insert(d_boxa[s].end(), sl.h_audio + (size_t)s * d_acap, sl.h_audio + (size_t)s * d_acap + c[1]);
            continue;
        }
        // gr_bit_sink::work (src/gr/gr_bit_sink.cpp:61-83): while more than 1 Mi items wait, new ones are not taken (nothing is cleared);
        // gr_const_sink::work (src/gr/gr_const_sink.cpp:64-86): the same at 256 items
        if (sl.bits_copied) {
            if (d_box1[s].size() <= 1048576) d_box1[s].insert(d_box1[s].end(), sl.h_a + (size_t)s * d_bcap, sl.h_a + (size_t)s * d_bcap + c[2]);
            if (d_box2[s].size() <= 1048576) d_box2[s].insert(d_box2[s].end(), sl.h_b + (size_t)s * d_bcap, sl.h_b + (size_t)s * d_bcap + c[3]);
        }
        if (sl.framed) for (int k = 0; k < 2; ++k) {   // the records of this call, in order
            d_fbits[k][s] += c[2 + k];
            d_fact[k][s] += sl.h_frcnt[k][2 * (size_t)d_n + s];
            const uint8_t* p = sl.h_fr[k] + (size_t)s * d_frcap;
            const uint32_t nbytes = std::min<uint32_t>(sl.h_frcnt[k][2 * s], (uint32_t)d_frcap);
            for (uint32_t pos = 0; pos + 8 <= nbytes;) {
                uint32_t hdr[2];
                std::memcpy(hdr, p + pos, 8);
                const uint32_t nb = hdr[1] & 0xFFFFu, padded = (nb + 3u) & ~3u;
                if (pos + 8 + padded > nbytes) break;
                frame_record r;
                r.type = hdr[0]; r.modem_sync = hdr[1] >> 16;
                r.payload.assign(p + pos + 8, p + pos + 8 + padded);
                r.payload.resize((size_t)padded + 4, 0);   // processReceivedData reads frame_length + 1 bytes
                d_boxf[k][s].push_back(std::move(r));
                pos += 8 + padded;
            }
        }
        if (sl.const_copied && d_boxc[s].size() <= 256) d_boxc[s].insert(d_boxc[s].end(), sl.h_const + (size_t)s * d_ccap, sl.h_const + (size_t)s * d_ccap + c[1]);
        if (d_mode == QRL_MODEM_DMR) {
            if (sl.h_dmocnt[s] > kDmoCap) d_dmo_dropped += sl.h_dmocnt[s] - kDmoCap;   // more bursts in one call than the record buffer holds
            for (uint32_t i = 0; i < sl.h_dmocnt[s] && i < kDmoCap; ++i) {
                const uint8_t* r = sl.h_dmo + ((size_t)s * kDmoCap + i) * QRL_DMO_RECORD_BYTES;
                d_boxd[s].emplace_back(r, r + QRL_DMO_RECORD_BYTES);
            }
        }
    }
}
void gr_demod_base_hip::flush()
{
    std::lock_guard<std::recursive_mutex> hg(d_hmutex);
    if (d_inflight >= 0) { harvest(d_inflight); d_inflight = -1; }
}
std::vector<unsigned char>* gr_demod_base_hip::getData(int nr, int stream)   // gr_bit_sink::get_data: >= 32 bits or nothing (src/gr/gr_bit_sink.cpp:45-59)
{
    std::lock_guard<std::mutex> g(d_mutex);
    std::vector<unsigned char>& box = nr == 2 ? d_box2[stream] : d_box1[stream];
    if (box.size() < 32) return nullptr;
    std::vector<unsigned char>* out = new std::vector<unsigned char>;
    out->swap(box);
    return out;
}
std::vector<float>* gr_demod_base_hip::getAudio(int stream)   // gr_demod_base::getAudio (src/gr/gr_demod_base.cpp:968-976)
{
    // gr_audio_sink::get_data (src/gr/gr_audio_sink.cpp:51-66): packets of 640 samples (40 ms at 8 ksps at the least), else nothing
    std::lock_guard<std::mutex> g(d_mutex);
    std::vector<float>& box = d_boxa[stream];
    if (box.size() < 640) return nullptr;
    std::vector<float>* out = new std::vector<float>(box.begin(), box.begin() + 640);
    box.erase(box.begin(), box.begin() + 640);
    return out;
}
void gr_demod_base_hip::set_squelch(int value)   // gr_demod_base.cpp:1186-1199
{
    std::lock_guard<std::recursive_mutex> hg(d_hmutex);
    d_squelch = value;
    if (d_h && d_acap) chk(qrl_demod_set_squelch(d_h, (double)value), "qrl_demod_set_squelch");
}
void gr_demod_base_hip::set_ctcss(float value)   // gr_demod_base.cpp:1212-1218 -> gr_demod_nbfm::set_ctcss on both NBFM instances (gr_demod_nbfm.cpp:97-123)
{
    std::lock_guard<std::recursive_mutex> hg(d_hmutex);
    d_ctcss = value;
    if (d_h && d_acap && (d_mode == QRL_MODEM_NBFM2500 || d_mode == QRL_MODEM_NBFM5000)) chk(qrl_demod_set_ctcss(d_h, value), "qrl_demod_set_ctcss");
}
void gr_demod_base_hip::set_filter_width(int filter_width, int mode)   // gr_demod_base.cpp:1155-1185
{
    std::lock_guard<std::recursive_mutex> hg(d_hmutex);
    switch (mode) {
    case QRL_MODEM_WBFM: case QRL_MODEM_AM5000: case QRL_MODEM_NBFM2500: case QRL_MODEM_NBFM5000: case QRL_MODEM_USB2500: case QRL_MODEM_LSB2500: break;
    default: return;   // the reference's default branch
    }
    if (d_h && d_acap && d_mode == mode) {
        flush();   // the chain restarts: what the running call produced belongs to the old filters (recursive lock)
        chk(qrl_demod_set_filter_width(d_h, filter_width), "qrl_demod_set_filter_width");
    }
    d_width[mode] = filter_width;
}
void gr_demod_base_hip::set_gain(float value)   // gr_demod_base.cpp:1206-1210 -> gr_demod_ssb::set_gain on both SSB instances
{
    std::lock_guard<std::recursive_mutex> hg(d_hmutex);
    d_if_gain = value;
    if (d_h && d_acap && (d_mode == QRL_MODEM_USB2500 || d_mode == QRL_MODEM_LSB2500)) chk(qrl_demod_set_gain(d_h, value), "qrl_demod_set_gain");
}
void gr_demod_base_hip::set_agc_attack(float value)   // :1428-1448
{
    std::lock_guard<std::recursive_mutex> hg(d_hmutex);
    d_agc_attack = value;
    if (d_h && d_acap && d_mode == QRL_MODEM_AM5000) chk(qrl_demod_set_agc(d_h, d_agc_attack, d_agc_decay), "qrl_demod_set_agc");
}
void gr_demod_base_hip::set_agc_decay(float value)   // :1450-1470
{
    std::lock_guard<std::recursive_mutex> hg(d_hmutex);
    d_agc_decay = value;
    if (d_h && d_acap && d_mode == QRL_MODEM_AM5000) chk(qrl_demod_set_agc(d_h, d_agc_attack, d_agc_decay), "qrl_demod_set_agc");
}
std::vector<gr_complex>* gr_demod_base_hip::get_constellation_data(int stream)
{
    std::lock_guard<std::mutex> g(d_mutex);
    if (d_boxc[stream].size() < 32) return nullptr;      // gr_const_sink::get_data (src/gr/gr_const_sink.cpp:48-62)
    std::vector<gr_complex>* out = new std::vector<gr_complex>;
    out->swap(d_boxc[stream]);
    return out;
}
float gr_demod_base_hip::get_rssi(int stream)   // gr_demod_base.cpp:1234-1237
{
    std::lock_guard<std::mutex> g(d_mutex);
    return d_level[stream];
}
void gr_demod_base_hip::calibrate_rssi(float value)   // :1413-1418
{
    std::lock_guard<std::recursive_mutex> hg(d_hmutex);
    d_rssi_cal = value;
    if (d_rssi) chk(qrl_rssi_set_level(d_rssi, value), "qrl_rssi_set_level");
}
void gr_demod_base_hip::enable_time_domain(bool value)   // gr_demod_base.cpp:1115-1147 (+ gr_sample_sink::set_enabled)
{
    std::lock_guard<std::recursive_mutex> hg(d_hmutex);
    d_scope_on = value;
}
void gr_demod_base_hip::reconfigure_scope(int samp_rate, double filter_width)
{
    // The engine plans the scope decimator at create time and is stricter than the reference's setters (rates above 500 ksps, widths above 500 kHz and
    // filters of more than 4096 taps -- rates below ~2.4 ksps, widths below ~590 Hz -- are rejected).  A rejected value must not cost the caller the
    // demodulator: the old settings come back, the handle is re-opened with them, and the call throws std::invalid_argument.
    const int old_rate = d_scope_rate; const double old_fw = d_scope_fw;
    if (samp_rate > 500000 || filter_width < 0.0 || filter_width > 500000.0 || (samp_rate > 0 && samp_rate / 2 - samp_rate / 8 <= 0))
        throw std::invalid_argument("gr_demod_base_hip: time-domain sample rate / filter width outside what the engine plans (rate <= 500000, width <= 500000)");
    d_scope_rate = samp_rate; d_scope_fw = filter_width;
    if (d_mode < 0) return;
    flush();
    try { open(); }
    catch (const std::exception& e) {
        d_scope_rate = old_rate; d_scope_fw = old_fw;
        open();   // the settings the handle had before: this worked when it was created
        throw std::invalid_argument(std::string("gr_demod_base_hip: time-domain settings rejected, previous ones restored: ") + e.what());
    }
    std::lock_guard<std::mutex> g(d_mutex);
    for (auto& b : d_boxs) b.clear();   // the handle keeps history for the filter it was created with
}
void gr_demod_base_hip::set_time_sink_samp_rate(int samp_rate)   // gr_demod_base.cpp:1249-1290: a new resampler (decimation 1e6 / samp_rate, its own low-pass) under lock()
{
    std::lock_guard<std::recursive_mutex> hg(d_hmutex);
    if ((unsigned)samp_rate > 1000000u) return;   // :1251-1252
    reconfigure_scope(samp_rate, 0.0);
}
void gr_demod_base_hip::set_time_domain_filter_width(double filter_width)   // :1292-1301: set_taps(low_pass(1, 1e6, width, width, HAMMING)) on the current resampler
{
    std::lock_guard<std::recursive_mutex> hg(d_hmutex);
    reconfigure_scope(d_scope_rate, filter_width);
}
void gr_demod_base_hip::set_sample_window(unsigned int size)   // gr_sample_sink::set_sample_window (:35-41): odd sizes go up by one
{
    std::lock_guard<std::mutex> g(d_mutex);
    if (size % 2 != 0) size = size + 1;
    d_window = size;
}
void gr_demod_base_hip::get_sample_data(float* sample_data, unsigned int& size, int stream)   // :988-1013 over gr_sample_sink::get_data (:49-66)
{
    std::lock_guard<std::mutex> g(d_mutex);
    std::vector<gr_complex>& box = d_boxs[stream];
    size = 0;
    if (box.size() < 2) return;
    unsigned int n = (unsigned int)std::min<size_t>(box.size(), d_window);
    if (n % 2 != 0) n = n - 1;
    // the reference writes the reals to [0, n) and the imaginaries to [n + 1, 2 n + 1) -- one slot is skipped (i + j + 1 with i == n, :1004-1009)
    for (unsigned int i = 0; i < n; i++) sample_data[i] = box[i].real();
    for (unsigned int j = 0; j < n; j++) sample_data[n + j + 1] = box[j].imag();
    size = n * 2;
    box.erase(box.begin(), box.begin() + n);
}
void gr_demod_base_hip::enable_gui_fft(bool value)   // :1110-1113
{
    std::lock_guard<std::recursive_mutex> hg(d_hmutex);
    d_fft_on = value;
    if (d_fft) chk(qrl_fft_set_enabled(d_fft, value ? 1 : 0), "qrl_fft_set_enabled");
}
void gr_demod_base_hip::set_fft_size(int size)   // :1227-1232
{
    std::lock_guard<std::recursive_mutex> hg(d_hmutex);
    flush();
    d_fftsize = (unsigned)size;
    if (d_fft) chk(qrl_fft_set_fft_size(d_fft, d_fftsize), "qrl_fft_set_fft_size");
    if (d_fftout) { (void)hipFree(d_fftout); d_fftout = nullptr; }
}
void gr_demod_base_hip::get_FFT_data(float* fft_data, unsigned int& fftSize, int stream)   // :978-986 -> rx_fft_c::get_fft_data
{
    fftSize = 0;
    std::lock_guard<std::recursive_mutex> hg(d_hmutex);
    if (!d_fft) return;
    if (!d_fftout) hchk(hipMalloc(reinterpret_cast<void**>(&d_fftout), (size_t)d_n * d_fftsize * sizeof(float)), "hipMalloc");
    unsigned got = 0;
    chk(qrl_fft_get_fft_data(d_fft, d_fftout, d_fftsize, &got), "qrl_fft_get_fft_data");
    if (!got) return;
    hipStream_t cs = static_cast<hipStream_t>(d_copy);
    d_fftlast.resize((size_t)d_n * got);   // every stream's spectrum of this frame: the other rows stay readable through last_FFT_data()
    hchk(hipMemcpyAsync(d_fftlast.data(), d_fftout, d_fftlast.size() * sizeof(float), hipMemcpyDeviceToHost, cs), "D2H");
    hchk(hipStreamSynchronize(cs), "hipStreamSynchronize");
    std::memcpy(fft_data, d_fftlast.data() + (size_t)stream * got, (size_t)got * sizeof(float));
    fftSize = got;
}
std::vector<std::vector<unsigned char>> gr_demod_base_hip::getDMRData(int stream)
{
    std::lock_guard<std::mutex> g(d_mutex);
    std::vector<std::vector<unsigned char>> out;
    out.swap(d_boxd[stream]);
    return out;
}

// ================================================================================================ gr_mod_base_hip
gr_mod_base_hip::gr_mod_base_hip(qrl_runtime& rt, int streams, int device_samp_rate, double carrier_offset_hz, size_t max_bytes)
    : d_rt(rt), d_n(streams), d_rate(device_samp_rate), d_offset(carrier_offset_hz), d_max(max_bytes), d_queue(streams), d_aqueue(streams), d_sent(streams, 0)
{
    d_cw_n = std::min<size_t>(1024, d_max);   // a facade built for fewer than 1024 items per call must still key CW
}
gr_mod_base_hip::~gr_mod_base_hip()
{
    if (d_h) qrl_mod_destroy(d_h);
    if (d_ah) qrl_amod_destroy(d_ah);
    if (d_bytes) (void)hipFree(d_bytes);
    if (d_audio) (void)hipFree(d_audio);
    if (d_iq) (void)hipFree(d_iq);
}
static bool analog_tx_mode(int mode)
{
    return mode == QRL_MODEM_NBFM2500 || mode == QRL_MODEM_NBFM5000 || mode == QRL_MODEM_AM5000 || mode == QRL_MODEM_USB2500 || mode == QRL_MODEM_LSB2500 ||
           mode == QRL_MODEM_CW600USB;
}
void gr_mod_base_hip::open()
{
    // create first, swap in on success: a mode / offset the engine rejects leaves the running modulator intact
    const bool backend = d_rate >= 2000000 || d_offset != 0.0;
    qrl_mod* nh = nullptr; qrl_amod* nah = nullptr; uint8_t* nbytes = nullptr; float* naudio = nullptr; float* niq = nullptr;
    size_t spblock = 0, bpb = 1;
    try {
        if (analog_tx_mode(d_mode)) {
            qrl_amod_config c{};
            c.modem_type = d_mode; c.batch = d_n; c.max_samples = d_max; c.bb_gain = d_gain;
            c.device_samp_rate = d_rate; c.carrier_offset_hz = d_offset;
            chk(qrl_amod_create(d_rt.ctx(), &c, &nah), "qrl_amod_create");
            // the reference's instances keep what their setters did across mode changes
            if (d_ctcss_touched && (d_mode == QRL_MODEM_NBFM2500 || d_mode == QRL_MODEM_NBFM5000)) chk(qrl_amod_set_ctcss(nah, d_ctcss), "qrl_amod_set_ctcss");
            if (d_width.count(d_mode)) chk(qrl_amod_set_filter_width(nah, d_width[d_mode]), "qrl_amod_set_filter_width");
            if (d_mode == QRL_MODEM_CW600USB && d_cw_key) chk(qrl_amod_set_cw_k(nah, 1), "qrl_amod_set_cw_k");
            hchk(hipMalloc(reinterpret_cast<void**>(&naudio), (size_t)d_n * d_max * sizeof(float)), "hipMalloc");
            hchk(hipMalloc(reinterpret_cast<void**>(&niq), (size_t)d_n * qrl_amod_out_cap(nah, d_max) * sizeof(gr_complex)), "hipMalloc");
        } else {
            qrl_mod_config c{};
            c.modem_type = d_mode; c.use_mode_defaults = 1; c.batch = d_n; c.max_bytes = d_max; c.bb_gain = d_gain;
            c.device_samp_rate = d_rate; c.carrier_offset_hz = d_offset;
            chk(qrl_mod_create(d_rt.ctx(), &c, &nh), "qrl_mod_create");
            hchk(hipMalloc(reinterpret_cast<void**>(&nbytes), (size_t)d_n * d_max), "hipMalloc");
            spblock = qrl_mod_samples_per_block(nh, &bpb);     // M17: 2500 samples per 3 bytes; every other mode: samples per byte, 1
            hchk(hipMalloc(reinterpret_cast<void**>(&niq), (size_t)d_n * (d_max / bpb + 1) * spblock * sizeof(gr_complex)), "hipMalloc");
        }
    } catch (...) {
        if (nh) qrl_mod_destroy(nh);
        if (nah) qrl_amod_destroy(nah);
        if (nbytes) (void)hipFree(nbytes);
        if (naudio) (void)hipFree(naudio);
        if (niq) (void)hipFree(niq);
        throw;
    }
    if (d_h) qrl_mod_destroy(d_h);
    if (d_ah) qrl_amod_destroy(d_ah);
    if (d_bytes) (void)hipFree(d_bytes);
    if (d_audio) (void)hipFree(d_audio);
    if (d_iq) (void)hipFree(d_iq);
    d_h = nh; d_ah = nah; d_bytes = nbytes; d_audio = naudio; d_iq = niq; d_backend = backend;
    if (nh) { d_spblock = spblock; d_bpb = bpb; }
    d_cw_n = std::min(d_cw_n, d_max);
    // a new handle counts its items from zero: whatever was queued for the old one (and the byte positions of its zero runs) goes with it
    std::lock_guard<std::mutex> g(d_mutex);
    for (auto& q : d_queue) q.clear();
    for (auto& q : d_aqueue) q.clear();
    for (auto& v : d_sent) v = 0;
}
void gr_mod_base_hip::set_mode(int mode)   // gr_mod_base::set_mode (src/gr/gr_mod_base.cpp:354-763): new graph, queued bytes dropped
{
    std::lock_guard<std::recursive_mutex> hg(d_hmutex);
    const int old = d_mode;
    d_mode = mode;
    try { open(); }
    catch (...) { d_mode = old; throw; }   // the previous modulator is still open
}
int gr_mod_base_hip::setDMRData(const std::vector<std::vector<uint8_t>>& frames, int stream)   // gr_mod_base.cpp:788-791 -> gr_dmr_source.cpp:56-73
{
    // d_hmutex is held by work() across qrl_mod_process: the bytes and the zero runs that belong to them become visible to a pass together
    // (gr_dmr_source::set_data takes the source's mutex for the same reason, gr_dmr_source.cpp:56-73)
    std::lock_guard<std::recursive_mutex> hg(d_hmutex);
    if (d_mode != QRL_MODEM_DMR || !d_h) throw std::runtime_error("gr_mod_base_hip::setDMRData outside DMR mode");
    constexpr size_t kZeroBytes = 33 + 2 * 3;            // DMR_ZERO_TX_LENGTH_BYTES = FRAME_LENGTH_BYTES + 2 CACH_LENGTH_BYTES (gr_dmr_source.cpp:24)
    std::vector<qrl_zero_run> runs;
    std::lock_guard<std::mutex> g(d_mutex);
    auto& q = d_queue[stream];
    const size_t q0 = q.size();
    for (const auto& f : frames) {
        q.insert(q.end(), f.begin(), f.end());
        const uint64_t first_zero = d_sent[stream] + q.size();   // byte index of the tag (gr_dmr_source.cpp:118-121)
        q.insert(q.end(), kZeroBytes, (uint8_t)0);
        qrl_zero_run z{};
        z.stream = stream; z.channel = 0; z.start = first_zero * 20u; z.count = kZeroBytes * 4u * 5u;   // DMR_ZERO_TX_LENGTH_SAMPLES (:25)
        runs.push_back(z);
    }
    if (!runs.empty() && qrl_mod_add_zero_runs(d_h, runs.data(), runs.size()) != QRL_OK) {
        q.resize(q0);   // bytes without their idle zeros would be a different waveform: all or nothing
        throw std::runtime_error(std::string("qrl_mod_add_zero_runs: ") + qrl_last_error());
    }
    return 0;
}
int gr_mod_base_hip::set_audio(std::vector<float>* data, int stream)   // gr_mod_base.cpp:793-797 -> gr_audio_source::set_data (gr_audio_source.cpp:55-66)
{
    std::lock_guard<std::mutex> g(d_mutex);
    d_aqueue[stream].insert(d_aqueue[stream].end(), data->begin(), data->end());
    delete data;
    return 0;
}
void gr_mod_base_hip::set_cw_k(bool value)   // gr_mod_base.cpp:948-956
{
    std::lock_guard<std::recursive_mutex> hg(d_hmutex);
    d_cw_key = value;
    if (d_ah && d_mode == QRL_MODEM_CW600USB) chk(qrl_amod_set_cw_k(d_ah, value ? 1 : 0), "qrl_amod_set_cw_k");
}
void gr_mod_base_hip::set_ctcss(float value)   // gr_mod_base.cpp:872-877
{
    std::lock_guard<std::recursive_mutex> hg(d_hmutex);
    d_ctcss = value; d_ctcss_touched = true;
    if (d_ah && (d_mode == QRL_MODEM_NBFM2500 || d_mode == QRL_MODEM_NBFM5000)) chk(qrl_amod_set_ctcss(d_ah, value), "qrl_amod_set_ctcss");
}
void gr_mod_base_hip::set_filter_width(int filter_width, int mode)   // gr_mod_base.cpp:878-905
{
    if (!analog_tx_mode(mode)) return;   // the reference's default branch
    std::lock_guard<std::recursive_mutex> hg(d_hmutex);
    if (d_ah && d_mode == mode) chk(qrl_amod_set_filter_width(d_ah, filter_width), "qrl_amod_set_filter_width");
    d_width[mode] = filter_width;
}
int gr_mod_base_hip::set_data(std::vector<uint8_t>* data, int stream)   // gr_mod_base.cpp:783-786 -> gr_byte_source::set_data (:54-62)
{
    std::lock_guard<std::mutex> g(d_mutex);
    d_queue[stream].insert(d_queue[stream].end(), data->begin(), data->end());
    delete data;
    return 1;
}
void gr_mod_base_hip::set_bb_gain(float v)
{
    std::lock_guard<std::recursive_mutex> hg(d_hmutex);
    d_gain = v;
    if (d_h) chk(qrl_mod_set_bb_gain(d_h, v), "qrl_mod_set_bb_gain");
    if (d_ah) chk(qrl_amod_set_bb_gain(d_ah, v), "qrl_amod_set_bb_gain");
}
static bool tx_mode_has_back_end(int mode)   // qrl_mod_create builds the gr_mod_base back end (rotator + interpolator) for every family but these (tx.cpp)
{
    return !(mode == QRL_MODEM_M17 || mode == QRL_MODEM_BPSK8);
}
void gr_mod_base_hip::set_carrier_offset(double hz)
{
    std::lock_guard<std::recursive_mutex> hg(d_hmutex);
    // a handle opened at 1 Msps with zero offset has no back end (no rotator to retune): zero stays a no-op, the first non-zero offset re-opens the handle with one
    // (the modulator restarts and what was queued is dropped; the reference's rotator is always in the graph, gr_mod_base.cpp:38).  The new handle is created
    // BEFORE the old one goes (open()), so a mode whose back end is not built (M17, DSSS) keeps its modulator and its offset, and the call throws.
    if (!d_backend) {
        if (hz == d_offset) return;
        if (d_mode >= 0 && !tx_mode_has_back_end(d_mode))
            throw std::invalid_argument("gr_mod_base_hip::set_carrier_offset: this mode's modulator has no rotator (M17 / DSSS back end not built); offset unchanged");
        const double old = d_offset;
        d_offset = hz;
        if (d_mode >= 0) {
            try { open(); }
            catch (...) { d_offset = old; throw; }
        }
        return;
    }
    if (d_h) chk(qrl_mod_set_carrier_offset(d_h, hz), "qrl_mod_set_carrier_offset");
    if (d_ah) chk(qrl_amod_set_carrier_offset(d_ah, hz), "qrl_amod_set_carrier_offset");
    d_offset = hz;
}
void gr_mod_base_hip::set_samp_rate(int device_samp_rate)
{
    std::lock_guard<std::recursive_mutex> hg(d_hmutex);
    if (device_samp_rate == d_rate) return;
    const int old = d_rate;
    d_rate = device_samp_rate;
    if (d_mode >= 0) {
        try { open(); }
        catch (...) { d_rate = old; throw; }   // the previous modulator is still open
    }
}
void gr_mod_base_hip::flush_sources()
{
    std::lock_guard<std::recursive_mutex> hg(d_hmutex);
    std::lock_guard<std::mutex> g(d_mutex);
    // (bytes already counted into d_sent stay counted: the positions of later zero runs are stream positions of the modulator)
    for (auto& q : d_queue) q.clear();
    for (auto& q : d_aqueue) q.clear();
}
size_t gr_mod_base_hip::samples_per_byte() const { return d_h ? qrl_mod_samples_per_byte(d_h) : 0; }
size_t gr_mod_base_hip::work(gr_complex* const* out)
{
    std::lock_guard<std::recursive_mutex> hg(d_hmutex);   // the handle and its zero-run list for the whole pass; the byte / audio queues stay under d_mutex
    if (d_ah && d_mode == QRL_MODEM_CW600USB) {   // the tone source lives on the device: nothing to upload
        const size_t stride = qrl_amod_out_cap(d_ah, d_max);
        chk(qrl_amod_process(d_ah, nullptr, 0, d_cw_n, d_iq, stride), "qrl_amod_process");
        chk(qrl_amod_sync(d_ah), "qrl_amod_sync");
        const size_t ns = qrl_amod_last_count(d_ah);
        for (int s = 0; s < d_n && ns; ++s)
            hchk(hipMemcpy(out[s], d_iq + 2 * (size_t)s * stride, ns * sizeof(gr_complex), hipMemcpyDeviceToHost), "D2H");
        return ns;
    }
    if (d_ah) {
        size_t na = 0;
        std::vector<float> host((size_t)d_n * d_max, 0.0f);
        {
            std::lock_guard<std::mutex> g(d_mutex);
            for (auto& q : d_aqueue) na = std::max(na, std::min(q.size(), d_max));
            if (d_mode == QRL_MODEM_NBFM2500 || d_mode == QRL_MODEM_NBFM5000) na -= na % 4;   // the 25:4 resampler; the rest waits for the next call
            if (na == 0) return 0;
            for (int s = 0; s < d_n; ++s) {   // a stream with fewer samples queued sends silence
                const size_t k = std::min(d_aqueue[s].size(), na);
                std::memcpy(host.data() + (size_t)s * d_max, d_aqueue[s].data(), k * sizeof(float));
                d_aqueue[s].erase(d_aqueue[s].begin(), d_aqueue[s].begin() + k);
            }
        }
        hipStream_t ms = static_cast<hipStream_t>(qrl_amod_stream(d_ah));
        hchk(hipMemcpyAsync(d_audio, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice, ms), "H2D");
        const size_t stride = qrl_amod_out_cap(d_ah, d_max);
        chk(qrl_amod_process(d_ah, d_audio, d_max, na, d_iq, stride), "qrl_amod_process");
        chk(qrl_amod_sync(d_ah), "qrl_amod_sync");
        const size_t ns = qrl_amod_last_count(d_ah);
        for (int s = 0; s < d_n && ns; ++s)
            hchk(hipMemcpy(out[s], d_iq + 2 * (size_t)s * stride, ns * sizeof(gr_complex), hipMemcpyDeviceToHost), "D2H");
        return ns;
    }
    if (!d_h) throw std::runtime_error("gr_mod_base_hip::work before set_mode");
    size_t nb = 0;
    std::vector<uint8_t> host((size_t)d_n * d_max, 0);
    {
        std::lock_guard<std::mutex> g(d_mutex);
        for (auto& q : d_queue) nb = std::max(nb, std::min(q.size(), d_max));
        nb -= nb % d_bpb;                 // whole blocks only (M17: 3 bytes); the rest waits for the next call
        if (nb == 0) return 0;
        for (int s = 0; s < d_n; ++s) {   // a stream with fewer bytes queued sends zero bytes, like an idle gr_byte_source feeding zeros
            const size_t k = std::min(d_queue[s].size(), nb);
            std::memcpy(host.data() + (size_t)s * d_max, d_queue[s].data(), k);
            d_queue[s].erase(d_queue[s].begin(), d_queue[s].begin() + k);
            d_sent[s] += nb;
        }
    }
    hipStream_t ms = static_cast<hipStream_t>(qrl_mod_stream(d_h));
    hchk(hipMemcpyAsync(d_bytes, host.data(), host.size(), hipMemcpyHostToDevice, ms), "H2D");
    const size_t ns = nb / d_bpb * d_spblock, stride = (d_max / d_bpb + 1) * d_spblock;
    chk(qrl_mod_process(d_h, d_bytes, d_max, nb, d_iq, stride), "qrl_mod_process");
    chk(qrl_mod_sync(d_h), "qrl_mod_sync");
    for (int s = 0; s < d_n; ++s)
        hchk(hipMemcpy(out[s], d_iq + 2 * (size_t)s * stride, ns * sizeof(gr_complex), hipMemcpyDeviceToHost), "D2H");
    return ns;
}

// ================================================================================================ gr_modem_hip
gr_modem_hip::gr_modem_hip(gr_demod_base_hip* demod, gr_mod_base_hip* mod, gr_modem_events events)
    : _gr_demod_base(demod), _gr_mod_base(mod), _ev(std::move(events)), _rx(2 * (size_t)(demod ? demod->streams() : (mod ? mod->streams() : 1)))
{
}
void gr_modem_hip::toggleRxMode(int modem_type)   // gr_modem.cpp:203-322
{
    _modem_type_rx = modem_type;
    if (!_gr_demod_base) return;
    _gr_demod_base->enable_device_framing(_device_framing);
    _gr_demod_base->set_mode(modem_type);
    _rx_frame_length = modem_rx_frame_length(modem_type, &_bit_buf_len);
    for (auto& r : _rx) { r = rx_state(); r.bit_buf.assign((size_t)std::max(_bit_buf_len, 8), 0); }
}
void gr_modem_hip::toggleTxMode(int modem_type)   // gr_modem.cpp:105-199
{
    _modem_type_tx = modem_type;
    if (!_gr_mod_base) return;
    _gr_mod_base->set_mode(modem_type);
    _tx_frame_length = modem_tx_frame_length(modem_type);
}

bool gr_modem_hip::demodulateAnalog(int stream)
{
    if (!_gr_demod_base) return false;
    std::vector<float>* audio_data = _gr_demod_base->getAudio(stream);
    if (audio_data == nullptr) return false;
    if (audio_data->size() > 0) {
        if (_ev.pcmAudio) _ev.pcmAudio(stream, audio_data); else delete audio_data;
        return true;
    }
    delete audio_data;
    return false;
}
bool gr_modem_hip::demodulate(int stream)   // gr_modem.cpp:1019-1117
{
    if (!_gr_demod_base) return false;
    if (_modem_type_rx == QRL_MODEM_DMR) {   // :1025-1038 (getDMRData -> DMRControl::addFrames)
        auto frames = _gr_demod_base->getDMRData(stream);
        if (frames.empty()) return false;
        if (_ev.dmrFrames) _ev.dmrFrames(stream, frames);
        return true;
    }
    const bool two = modem_two_branches(_modem_type_rx);
    if (device_framing()) {
        // the synchroniser ran on the device (qrl_framesync_process behind the demodulator): what arrives here are the frames
        // gr_modem::synchronize would have cut out of the bit stream, with _modem_sync as it stood when each frame completed.
        // Same dispatch as the host loop (processReceivedData, :1285-1441); two-branch modes: branch A's frames, then branch B's
        // (BranchRuleReference: branch A only -- the `>=` rule of :1080-1090 with equal counts).
        // Return value = the reference's: getData() hands out nothing below 32 bits -- then demodulate() returns false and consumes
        // nothing (:1057-1071; two-branch modes need both vectors) -- else synchronize()'s data_to_process: true iff a bit of this batch was collected
        // while a sync was held (:1121-1175), which k_framesync counts per call (qrl_framesync_set_activity_output).
        if (_gr_demod_base->peekFrameBits(1, stream) < 32 || (two && _gr_demod_base->peekFrameBits(2, stream) < 32)) return false;
        bool data_to_process = false;
        for (int k = 0; k < (two ? 2 : 1); ++k) {
            std::vector<gr_demod_base_hip::frame_record> recs;
            size_t bits = 0, collected = 0;
            if (!_gr_demod_base->takeFrames(k + 1, stream, recs, bits, collected)) continue;
            if (k == 1 && _branch_rule != BranchRuleBoth) continue;   // BranchRuleReference: branch B's vector is fetched and dropped
            rx_state& r = _rx[2 * (size_t)stream + (size_t)k];
            data_to_process = data_to_process || collected > 0;
            for (auto& f : recs) {
                r.modem_sync = (int)f.modem_sync;
                r.current_frame_type = f.type;
                processReceivedData(f.payload.data(), f.type, r, stream);
            }
        }
        return data_to_process;
    }
    std::vector<unsigned char>*demod_data = nullptr, *demod_data2 = nullptr;
    if (two) {
        demod_data = _gr_demod_base->getData(1, stream);
        demod_data2 = _gr_demod_base->getData(2, stream);
        if (demod_data == nullptr || demod_data2 == nullptr) {
            // (the reference leaks whichever branch did come back, :1062-1063; here it is freed)
            delete demod_data; delete demod_data2;
            return false;
        }
    } else {
        demod_data = _gr_demod_base->getData(stream);
        if (demod_data == nullptr) return false;
    }
    bool data_to_process;
    if (two && _branch_rule == BranchRuleReference) {
        // :1080-1090: the longer vector wins, `>=` favours branch 1.  In the reference the two sinks are filled by scheduler
        // threads, so the comparison is an accident of timing; the device fills both mailboxes with the same count every call,
        // which turns the rule into "always branch 1" (kept selectable for like-for-like comparisons with the reference).
        std::vector<unsigned char>* data = demod_data->size() >= demod_data2->size() ? demod_data : demod_data2;
        data_to_process = synchronize((int)data->size(), data, _rx[2 * (size_t)stream], stream);
    } else if (two) {
        // default: both Viterbi alignments keep their own frame synchroniser; only the branch whose decoder is aligned with
        // the transmitter's code words ever finds a sync word, so frames are delivered once
        const bool a = synchronize((int)demod_data->size(), demod_data, _rx[2 * (size_t)stream], stream);
        const bool b = synchronize((int)demod_data2->size(), demod_data2, _rx[2 * (size_t)stream + 1], stream);
        data_to_process = a || b;
    } else {
        data_to_process = synchronize((int)demod_data->size(), demod_data, _rx[2 * (size_t)stream], stream);
    }
    delete demod_data;
    delete demod_data2;
    return data_to_process;
}

bool gr_modem_hip::synchronize(int v_size, std::vector<unsigned char>* data, rx_state& r, int stream)   // gr_modem.cpp:1119-1181
{
    bool data_to_process = false;
    const int m = _modem_type_rx;
    for (int i = 0; i < v_size; i++) {
        if (!r.sync_found) {
            r.current_frame_type = findSync((*data)[i], r);
            if (r.sync_found) {
                r.bit_buf_index = 0;
                if (r.modem_sync < 32) r.modem_sync += 8;
                continue;
            } else if (r.modem_sync > 0) {
                r.modem_sync -= 1;
            }
        }
        if (r.sync_found) {
            data_to_process = true;
            r.bit_buf[r.bit_buf_index] = (*data)[i] & 0x1;
            r.bit_buf_index++;
            int frame_length = _rx_frame_length, bit_buf_len = _bit_buf_len;
            if (!is_1k(m) && m != ModemTypeM17 && r.current_frame_type == FrameTypeVoice) frame_length++;   // reserved data
            else if (!is_1k(m) && m != ModemTypeM17 && r.current_frame_type != FrameTypeVoice) bit_buf_len = _bit_buf_len - 8;
            if (r.bit_buf_index >= bit_buf_len) {
                std::vector<unsigned char> frame_data((size_t)frame_length + 1, 0);
                for (int k = 0; k + 8 <= bit_buf_len; k += 8) {   // packBytes, :980-994
                    int t = 0;
                    for (int b = 0; b < 8; ++b) t = (t << 1) | (r.bit_buf[k + b] & 0x1);
                    if (k / 8 < (int)frame_data.size()) frame_data[k / 8] = (unsigned char)t;
                }
                processReceivedData(frame_data.data(), r.current_frame_type, r, stream);
                r.sync_found = false;
                r.shift_reg = 0;
                r.bit_buf_index = 0;
            }
        }
    }
    return data_to_process;
}

uint64_t gr_modem_hip::findSync(unsigned char bit, rx_state& r)   // gr_modem.cpp:1183-1282
{
    r.shift_reg = (r.shift_reg << 1) | (bit & 0x1);
    uint64_t temp;
    const int m = _modem_type_rx;
    if (m == ModemTypeM17) {   // :1187-1210
        temp = r.shift_reg & 0xFFFF;
        if (temp == FrameTypeM17LSF) { r.sync_found = true; return FrameTypeM17LSF; }
        if (temp == FrameTypeM17Stream) { r.sync_found = true; return FrameTypeM17Stream; }
        temp = r.shift_reg & 0xFFFFFFFF;
        if (temp == FrameTypeM17EOT) { r.sync_found = true; return FrameTypeM17EOT; }
        return FrameTypeNone;
    }
    if (is_1k(m)) {
        temp = r.shift_reg & 0xFF;
        if (temp == FrameTypeVoice1) { r.sync_found = true; return FrameTypeVoice1; }
        return FrameTypeNone;
    }
    if (m != QRL_MODEM_QPSK250K && m != QRL_MODEM_QPSKVIDEO && m != QRL_MODEM_4FSK100K) {
        temp = r.shift_reg & 0xFFFF;
        if (temp == FrameTypeVoice2) { r.sync_found = true; return FrameTypeVoice; }
        temp = r.shift_reg & 0xFFFFFF;
        if (temp == FrameTypeText) { r.sync_found = true; return FrameTypeText; }
        if (temp == FrameTypeProto) { r.sync_found = true; return FrameTypeProto; }
        if (temp == FrameTypeVideo) { r.sync_found = true; return FrameTypeVideo; }
        if (temp == FrameTypeCallsign) { r.sync_found = true; return FrameTypeCallsign; }
        if (temp == FrameTypeEnd) { r.sync_found = true; return FrameTypeEnd; }
        return FrameTypeNone;
    }
    temp = r.shift_reg & 0xFFFFFF;
    if (temp == FrameTypeIP) { r.sync_found = true; return FrameTypeIP; }
    if (temp == FrameTypeVideo) { r.sync_found = true; return FrameTypeVideo; }
    if (temp == FrameTypeEnd) { r.sync_found = true; return FrameTypeEnd; }
    return FrameTypeNone;
}

void gr_modem_hip::processReceivedData(unsigned char* received_data, uint64_t current_frame_type, rx_state& r, int stream)   // gr_modem.cpp:1285-1441
{
    const int L = _rx_frame_length;
    if (current_frame_type == FrameTypeEnd) {
        handleStreamEnd(r, stream);
    } else if (current_frame_type == FrameTypeText) {
        if (_ev.dataFrameReceived) _ev.dataFrameReceived(stream);
        r.last_frame_type = FrameTypeText;
        int string_length = L;
        for (int ii = L - 1; ii >= 0; ii--) { if (received_data[ii] == 0) string_length--; else break; }   // trailing NULs off
        if (_ev.textReceived) _ev.textReceived(stream, std::string(reinterpret_cast<const char*>(received_data), (size_t)string_length), false);
    } else if (current_frame_type == FrameTypeProto) {
        if (_ev.dataFrameReceived) _ev.dataFrameReceived(stream);
        r.last_frame_type = FrameTypeProto;
        if (_ev.protoReceived) _ev.protoReceived(stream, std::vector<unsigned char>(received_data, received_data + L));
    } else if (current_frame_type == FrameTypeCallsign) {
        r.last_frame_type = FrameTypeCallsign;
        std::string callsign;
        for (int i = 0; i < 7 && received_data[i]; ++i) {   // QRegExp("[^a-zA-Z/\\d\\s]") removed, first 7 characters kept
            const unsigned char c = received_data[i];
            if ((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9') || c == '/' || c == ' ' || (c >= 9 && c <= 13)) callsign.push_back((char)c);
        }
        if (_ev.callsignReceived) _ev.callsignReceived(stream, callsign);
    } else if (current_frame_type == FrameTypeVoice1) {
        r.last_frame_type = FrameTypeVoice1;
        if (r.modem_sync >= 16 && _ev.digitalAudio) _ev.digitalAudio(stream, received_data, L);   // the voice gate of the 1k modes
    } else if (current_frame_type == FrameTypeVoice2) {
        r.last_frame_type = FrameTypeVoice2;
        if (_ev.digitalAudio) _ev.digitalAudio(stream, received_data + 1, L);   // one reserved byte in front
    } else if (current_frame_type == FrameTypeVideo) {
        if (_ev.dataFrameReceived) _ev.dataFrameReceived(stream);
        r.last_frame_type = FrameTypeVideo;
        if (_ev.videoData) _ev.videoData(stream, received_data, L);
    } else if (current_frame_type == FrameTypeIP) {
        r.last_frame_type = FrameTypeIP;
        if (_ev.netData) _ev.netData(stream, received_data, L);
    } else if (current_frame_type == FrameTypeM17Stream || current_frame_type == FrameTypeM17LSF || current_frame_type == FrameTypeM17EOT) {
        if (_ev.m17Frame) _ev.m17Frame(stream, current_frame_type, received_data, L);   // the M17 decoder itself is out of scope
    }
}
void gr_modem_hip::handleStreamEnd(rx_state& r, int stream)   // gr_modem.cpp:1443-1451
{
    if (r.last_frame_type == FrameTypeText && _ev.textReceived) _ev.textReceived(stream, "\n", false);
    if (_ev.endAudioTransmission) _ev.endAudioTransmission(stream);
    if (_ev.receiveEnd) _ev.receiveEnd(stream);
}

// ---- TX
std::vector<unsigned char>* gr_modem_hip::frame(unsigned char* encoded_audio, int data_size, int frame_type)   // gr_modem.cpp:904-961
{
    std::vector<unsigned char>* data = new std::vector<unsigned char>;
    if ((uint64_t)frame_type == FrameTypeIP && _burst_ip_modem) for (int i = 0; i < 10; i++) data->push_back(0xAA);
    auto push3 = [&](uint64_t t) { data->push_back((unsigned char)((t >> 16) & 0xFF)); data->push_back((unsigned char)((t >> 8) & 0xFF)); data->push_back((unsigned char)(t & 0xFF)); };
    if ((uint64_t)frame_type == FrameTypeVoice) {
        if (is_1k(_modem_type_tx)) data->push_back((unsigned char)(FrameTypeVoice1 & 0xFF));
        else { data->push_back((unsigned char)((FrameTypeVoice2 >> 8) & 0xFF)); data->push_back((unsigned char)(FrameTypeVoice2 & 0xFF)); data->push_back(0xAA); }
    } else if ((uint64_t)frame_type == FrameTypeText) push3(FrameTypeText);
    else if ((uint64_t)frame_type == FrameTypeVideo) push3(FrameTypeVideo);
    else if ((uint64_t)frame_type == FrameTypeIP) push3(FrameTypeIP);
    else if ((uint64_t)frame_type == FrameTypeProto) push3(FrameTypeProto);
    for (int i = 0; i < data_size; i++) data->push_back(encoded_audio[i]);
    return data;
}
void gr_modem_hip::transmit(std::vector<std::vector<unsigned char>*> frames, int stream)   // gr_modem.cpp:963-978
{
    if (!_gr_mod_base) { for (auto* f : frames) delete f; return; }
    std::vector<unsigned char>* all_frames = new std::vector<unsigned char>;
    for (auto* f : frames) { all_frames->insert(all_frames->end(), f->begin(), f->end()); delete f; }
    _gr_mod_base->set_data(all_frames, stream);
}
void gr_modem_hip::sendCallsign(const std::string& callsign, int stream)   // gr_modem.cpp:654-676
{
    std::vector<unsigned char>* f = new std::vector<unsigned char>;
    f->push_back((unsigned char)((FrameTypeCallsign >> 16) & 0xFF));
    f->push_back((unsigned char)((FrameTypeCallsign >> 8) & 0xFF));
    f->push_back((unsigned char)(FrameTypeCallsign & 0xFF));
    for (char c : callsign) f->push_back((unsigned char)c);
    for (int i = 0; i < _tx_frame_length - (int)callsign.size(); i++) f->push_back(0x00);
    transmit({f}, stream);
}
void gr_modem_hip::startTransmission(const std::string& callsign, int stream)   // gr_modem.cpp:678-707 (digital modes)
{
    if (!_gr_mod_base) return;
    std::vector<unsigned char>* tx_start = new std::vector<unsigned char>;
    for (int i = 0; i < 8; i++) tx_start->push_back(0xAA);   // preamble of 48 bits incl. ramp-up
    transmit({tx_start}, stream);
    sendCallsign(callsign, stream);
}
void gr_modem_hip::endTransmission(const std::string& callsign, int stream)   // gr_modem.cpp:710-744 (digital modes)
{
    std::vector<unsigned char>* tx_end = new std::vector<unsigned char>;
    _frame_counter = 0;
    sendCallsign(callsign, stream);
    tx_end->push_back((unsigned char)((FrameTypeEnd >> 16) & 0xFF));
    tx_end->push_back((unsigned char)((FrameTypeEnd >> 8) & 0xFF));
    tx_end->push_back((unsigned char)(FrameTypeEnd & 0xFF));
    for (int i = 0; i < _tx_frame_length * 10; i++) tx_end->push_back(0xAA);
    transmit({tx_end}, stream);
}
void gr_modem_hip::transmitDigitalAudio(unsigned char* data, int size, int stream) { transmit({frame(data, size, (int)FrameTypeVoice)}, stream); delete[] data; }   // :857-864
void gr_modem_hip::transmitVideoData(unsigned char* data, int size, int stream) { transmit({frame(data, size, (int)FrameTypeVideo)}, stream); delete[] data; }      // :892-899
void gr_modem_hip::transmitNetData(unsigned char* data, int size, int stream) { transmit({frame(data, size, (int)FrameTypeIP)}, stream); delete[] data; }           // :901-908
void gr_modem_hip::transmitTextData(const std::string& text, int frame_type, int stream)   // gr_modem.cpp:812-832
{
    std::vector<std::vector<unsigned char>*> frames;
    for (size_t k = 0; k < text.size(); k += (size_t)_tx_frame_length) {
        const std::string chunk = text.substr(k, (size_t)_tx_frame_length);
        std::vector<unsigned char> d((size_t)_tx_frame_length, 0);
        std::memcpy(d.data(), chunk.data(), chunk.size());
        frames.push_back(frame(d.data(), _tx_frame_length, frame_type));
    }
    transmit(frames, stream);
}
void gr_modem_hip::transmitBinData(const std::vector<unsigned char>& bin, int frame_type, int stream)   // gr_modem.cpp:834-855
{
    std::vector<std::vector<unsigned char>*> frames;
    for (size_t k = 0; k < bin.size(); k += (size_t)_tx_frame_length) {
        std::vector<unsigned char> d((size_t)_tx_frame_length, 0);
        std::memcpy(d.data(), bin.data() + k, std::min((size_t)_tx_frame_length, bin.size() - k));
        frames.push_back(frame(d.data(), _tx_frame_length, frame_type));
    }
    transmit(frames, stream);
}

}  // namespace qrl_host
