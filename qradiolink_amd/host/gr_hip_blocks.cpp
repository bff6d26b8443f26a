// gr_hip_blocks.cpp — see gr_hip_blocks.h.  Host buffers are staged through device memory with plain HIP
// copies on the handle's stream (a GNU Radio scheduler hands out host pointers).
#include "gr_hip_blocks.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <string>

static void chk(int rc, const char* what)
{
    if (rc != QRL_OK) throw std::runtime_error(std::string(what) + ": " + qrl_strerror(rc) + " (" + qrl_last_error() + ")");
}
static void hchk(hipError_t e, const char* what)
{
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

qrl_runtime::qrl_runtime(int device) { chk(qrl_init(device, &d_ctx), "qrl_init"); }
qrl_runtime::~qrl_runtime() { qrl_shutdown(d_ctx); }

enum { FAM_2FSK = 0, FAM_GMSK = 1, FAM_QPSK = 2, FAM_4FSK = 3, FAM_BPSK = 4, FAM_DMR = 5, FAM_M17 = 6, FAM_DSSS = 7, FAM_NBFM = 8, FAM_AM = 9, FAM_WBFM = 10, FAM_USB = 11, FAM_LSB = 12 };

gr_demod_hip_sptr make_gr_demod_2fsk_hip(qrl_runtime& rt, int sps, int samp_rate, int carrier_freq, int filter_width, bool fm)
{ return gr_demod_hip_sptr(new gr_demod_hip(rt, FAM_2FSK, sps, samp_rate, carrier_freq, filter_width, fm)); }
gr_demod_hip_sptr make_gr_demod_gmsk_hip(qrl_runtime& rt, int sps, int samp_rate, int carrier_freq, int filter_width)
{ return gr_demod_hip_sptr(new gr_demod_hip(rt, FAM_GMSK, sps, samp_rate, carrier_freq, filter_width, false)); }
gr_demod_hip_sptr make_gr_demod_qpsk_hip(qrl_runtime& rt, int sps, int samp_rate, int carrier_freq, int filter_width)
{ return gr_demod_hip_sptr(new gr_demod_hip(rt, FAM_QPSK, sps, samp_rate, carrier_freq, filter_width, false)); }

gr_demod_hip_sptr make_gr_demod_4fsk_hip(qrl_runtime& rt, int sps, int samp_rate, int carrier_freq, int filter_width, bool fm)
{ return gr_demod_hip_sptr(new gr_demod_hip(rt, FAM_4FSK, sps, samp_rate, carrier_freq, filter_width, fm)); }
gr_demod_hip_sptr make_gr_demod_bpsk_hip(qrl_runtime& rt, int sps, int samp_rate, int carrier_freq, int filter_width)
{ return gr_demod_hip_sptr(new gr_demod_hip(rt, FAM_BPSK, sps, samp_rate, carrier_freq, filter_width, false)); }
gr_demod_hip_sptr make_gr_demod_dmr_hip(qrl_runtime& rt, int sps, int samp_rate)
{ return gr_demod_hip_sptr(new gr_demod_hip(rt, FAM_DMR, sps, samp_rate, 1700, 5000, false)); }
gr_demod_hip_sptr make_gr_demod_m17_hip(qrl_runtime& rt, int sps, int samp_rate, int carrier_freq, int filter_width)
{ return gr_demod_hip_sptr(new gr_demod_hip(rt, FAM_M17, sps, samp_rate, carrier_freq, filter_width, false)); }

gr_demod_hip_sptr make_gr_demod_dsss_hip(qrl_runtime& rt, int sps, int samp_rate, int carrier_freq, int filter_width)
{ return gr_demod_hip_sptr(new gr_demod_hip(rt, FAM_DSSS, sps, samp_rate, carrier_freq, filter_width, false)); }

gr_demod_hip_sptr make_gr_demod_nbfm_hip(qrl_runtime& rt, int sps, int samp_rate, int carrier_freq, int filter_width)
{ return gr_demod_hip_sptr(new gr_demod_hip(rt, FAM_NBFM, sps, samp_rate, carrier_freq, filter_width, false)); }
gr_demod_hip_sptr make_gr_demod_am_hip(qrl_runtime& rt, int sps, int samp_rate, int carrier_freq, int filter_width)
{ return gr_demod_hip_sptr(new gr_demod_hip(rt, FAM_AM, sps, samp_rate, carrier_freq, filter_width, false)); }
gr_demod_hip_sptr make_gr_demod_wbfm_hip(qrl_runtime& rt, int sps, int samp_rate, int carrier_freq, int filter_width)
{ return gr_demod_hip_sptr(new gr_demod_hip(rt, FAM_WBFM, sps, samp_rate, carrier_freq, filter_width, false)); }

gr_demod_hip_sptr make_gr_demod_ssb_hip(qrl_runtime& rt, int sps, int samp_rate, int carrier_freq, int filter_width, int sb)
{ return gr_demod_hip_sptr(new gr_demod_hip(rt, sb ? FAM_LSB : FAM_USB, sps, samp_rate, carrier_freq, filter_width, false)); }

gr_demod_hip::gr_demod_hip(qrl_runtime& rt, int fam, int sps, int samp_rate, int carrier_freq, int filter_width, bool fm)
    : gr::sync_block("gr_demod_hip", gr::io_signature::make(1, 1, sizeof(gr_complex)), gr::io_signature::make(0, 0, 0)), d_rt(rt)
{
    // any member of the family selects the chain; the explicit factory arguments (not the mode table) configure it
    static const int rep[] = {QRL_MODEM_2FSK1K, QRL_MODEM_GMSK10K, QRL_MODEM_QPSK250K, QRL_MODEM_4FSK2KFM, QRL_MODEM_BPSK1K, QRL_MODEM_DMR, QRL_MODEM_M17, QRL_MODEM_BPSK8, QRL_MODEM_NBFM5000, QRL_MODEM_AM5000, QRL_MODEM_WBFM, QRL_MODEM_USB2500, QRL_MODEM_LSB2500};
    d_cfg.modem_type = rep[fam];
    d_cfg.use_mode_defaults = 0;
    d_cfg.sps = sps; d_cfg.samp_rate = samp_rate; d_cfg.carrier_freq = carrier_freq; d_cfg.filter_width = filter_width; d_cfg.fm = fm;
    d_cfg.device_samp_rate = samp_rate; d_cfg.carrier_offset_hz = 0.0;
    d_cfg.batch = 1; d_cfg.max_chunk = kChunk; d_cfg.enable_side_outputs = 1;
    open();
}
void gr_demod_hip::open()
{
    if (d_h) { qrl_demod_destroy(d_h); d_h = nullptr; }
    chk(qrl_demod_create(d_rt.ctx(), &d_cfg, &d_h), "qrl_demod_create");
    chk(qrl_demod_out_caps(d_h, kChunk, &d_fcap, &d_ccap, &d_bcap), "qrl_demod_out_caps");
    if (!d_iq) {
        hchk(hipMalloc(reinterpret_cast<void**>(&d_iq), kChunk * sizeof(gr_complex)), "hipMalloc");
        hchk(hipMalloc(reinterpret_cast<void**>(&d_cnt), 4 * sizeof(uint32_t)), "hipMalloc");
        hipStream_t cs;
        // lowest priority -- not for the scheduling: streams of one priority share a few hardware queues, and a wait queued on this
        // stream would otherwise hold back the kernels of a handle stream that happens to sit on the same queue (csrc/engine.cpp, stream creation)
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        hchk(hipStreamCreateWithPriority(&cs, hipStreamNonBlocking, prio_lo), "hipStreamCreate");
        d_cs = cs;
        hchk(hipHostMalloc(reinterpret_cast<void**>(&d_hcnt), 6 * sizeof(uint32_t), hipHostMallocDefault), "hipHostMalloc");
    }
    for (void* p : {(void*)d_const, (void*)d_a, (void*)d_b}) if (p) (void)hipFree(p);
    hchk(hipMalloc(reinterpret_cast<void**>(&d_const), d_ccap * sizeof(gr_complex)), "hipMalloc");
    hchk(hipMalloc(reinterpret_cast<void**>(&d_a), d_bcap), "hipMalloc");
    hchk(hipMalloc(reinterpret_cast<void**>(&d_b), d_bcap), "hipMalloc");
    d_ha.resize(d_bcap); d_hb.resize(d_bcap); d_hc.resize(d_ccap);
    chk(qrl_demod_audio_cap(d_h, kChunk, &d_acap), "qrl_demod_audio_cap");
    if (d_audio) { (void)hipFree(d_audio); d_audio = nullptr; }
    if (d_acap) hchk(hipMalloc(reinterpret_cast<void**>(&d_audio), d_acap * sizeof(float)), "hipMalloc");
    d_hau.resize(d_acap);
    if (d_df1) attach_deframer(d_df_type);   // the deframer buffers follow the (possibly new) bit capacity
}
gr_demod_hip::~gr_demod_hip()
{
    if (d_h) qrl_demod_destroy(d_h);
    if (d_df1) qrl_deframer_destroy(d_df1);
    if (d_df2) qrl_deframer_destroy(d_df2);
    for (void* p : {(void*)d_iq, (void*)d_const, (void*)d_a, (void*)d_b, (void*)d_cnt, (void*)d_fa, (void*)d_fb, (void*)d_fcnt, (void*)d_audio}) if (p) (void)hipFree(p);
    if (d_hcnt) (void)hipHostFree(d_hcnt);
    if (d_cs) (void)hipStreamDestroy(static_cast<hipStream_t>(d_cs));
}
void gr_demod_hip::attach_deframer(int type)
{
    if (d_df1) { qrl_deframer_destroy(d_df1); qrl_deframer_destroy(d_df2); d_df1 = d_df2 = nullptr; }
    d_df_type = type;
    chk(qrl_deframer_create(d_rt.ctx(), type, 1, d_cs, &d_df1), "qrl_deframer_create");   // on the copy stream, behind the demodulator
    chk(qrl_deframer_create(d_rt.ctx(), type, 1, d_cs, &d_df2), "qrl_deframer_create");
    d_dfcap = 2 * d_bcap + 24;
    for (void* p : {(void*)d_fa, (void*)d_fb, (void*)d_fcnt}) if (p) (void)hipFree(p);
    hchk(hipMalloc(reinterpret_cast<void**>(&d_fa), d_dfcap), "hipMalloc");
    hchk(hipMalloc(reinterpret_cast<void**>(&d_fb), d_dfcap), "hipMalloc");
    hchk(hipMalloc(reinterpret_cast<void**>(&d_fcnt), 2 * sizeof(uint32_t)), "hipMalloc");
    d_ha.resize(d_dfcap); d_hb.resize(d_dfcap);
}
void gr_demod_hip::set_device_samp_rate(int r) { d_cfg.device_samp_rate = r; open(); }
void gr_demod_hip::set_carrier_offset(double hz) { d_cfg.carrier_offset_hz = hz; chk(qrl_demod_set_carrier_offset(d_h, hz), "qrl_demod_set_carrier_offset"); }
void gr_demod_hip::flush()
{
    chk(qrl_demod_reset(d_h), "qrl_demod_reset");
    gr::thread::scoped_lock g(d_mutex);
    d_box1.clear(); d_box2.clear(); d_boxc.clear(); d_boxa.clear(); d_carry.clear();
}

void gr_demod_hip::run(const gr_complex* x, size_t n)   // n even, <= kChunk
{
    hipStream_t s = static_cast<hipStream_t>(qrl_demod_stream(d_h));
    hchk(hipMemcpyAsync(d_iq, x, n * sizeof(gr_complex), hipMemcpyHostToDevice, s), "H2D");
    qrl_demod_out o{};
    o.constellation = d_const; o.constellation_cap = d_ccap;
    o.bits_a = d_a; o.bits_b = d_b; o.bits_cap = d_bcap; o.counts = d_cnt;
    o.audio = d_audio; o.audio_cap = d_acap;
    chk(qrl_demod_process(d_h, d_iq, kChunk, n, &o), "qrl_demod_process");
    // everything behind the demodulator is chained on the copy stream with a device-side wait: the deframers (ports 2 / 3 ->
    // gr_deframer_bb on the device; the mailboxes then hold sync + frame bits) and the copy of the counts -- ONE host synchronisation
    // per call, after which the copies below move exactly the bytes the call produced
    hipStream_t cs = static_cast<hipStream_t>(d_cs);
    chk(qrl_demod_stream_wait(d_h, cs), "qrl_demod_stream_wait");
    if (d_df1) {
        chk(qrl_deframer_process(d_df1, d_a, d_bcap, d_bcap, d_cnt + 2, 4, d_fa, d_dfcap, d_fcnt), "qrl_deframer_process");
        chk(qrl_deframer_process(d_df2, d_b, d_bcap, d_bcap, d_cnt + 3, 4, d_fb, d_dfcap, d_fcnt + 1), "qrl_deframer_process");
        hchk(hipMemcpyAsync(d_hcnt + 4, d_fcnt, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, cs), "D2H");
    }
    hchk(hipMemcpyAsync(d_hcnt, d_cnt, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, cs), "D2H");
    hchk(hipStreamSynchronize(cs), "hipStreamSynchronize");
    uint32_t cnt[4] = {d_hcnt[0], d_hcnt[1], d_hcnt[2], d_hcnt[3]};
    if (d_df1) {
        cnt[2] = d_hcnt[4]; cnt[3] = d_hcnt[5];
        if (cnt[2]) hchk(hipMemcpy(d_ha.data(), d_fa, cnt[2], hipMemcpyDeviceToHost), "D2H");
        if (cnt[3]) hchk(hipMemcpy(d_hb.data(), d_fb, cnt[3], hipMemcpyDeviceToHost), "D2H");
    } else {
        if (cnt[2]) hchk(hipMemcpy(d_ha.data(), d_a, cnt[2], hipMemcpyDeviceToHost), "D2H");
        if (cnt[3]) hchk(hipMemcpy(d_hb.data(), d_b, cnt[3], hipMemcpyDeviceToHost), "D2H");
    }
    if (d_acap) {   // analogue modes: port 1 is audio
        if (cnt[1]) hchk(hipMemcpy(d_hau.data(), d_audio, cnt[1] * sizeof(float), hipMemcpyDeviceToHost), "D2H");
        gr::thread::scoped_lock g(d_mutex);
        if (d_boxa.size() > 8000) d_boxa.clear();      // gr_audio_sink::work: a backlog of more than one second is dropped (gr_audio_sink.cpp:79-85)
        else d_boxa.insert(d_boxa.end(), d_hau.begin(), d_hau.begin() + cnt[1]);
        return;
    }
    if (cnt[1]) hchk(hipMemcpy(d_hc.data(), d_const, cnt[1] * sizeof(gr_complex), hipMemcpyDeviceToHost), "D2H");
    gr::thread::scoped_lock g(d_mutex);
    if (d_box1.size() <= 1048576) d_box1.insert(d_box1.end(), d_ha.begin(), d_ha.begin() + cnt[2]);   // drop rule of gr_bit_sink.cpp:71-76
    if (d_box2.size() <= 1048576) d_box2.insert(d_box2.end(), d_hb.begin(), d_hb.begin() + cnt[3]);
    if (d_boxc.size() <= 256) d_boxc.insert(d_boxc.end(), d_hc.begin(), d_hc.begin() + cnt[1]);   // drop rule of gr_const_sink.cpp:75-78
}

int gr_demod_hip::work(int noutput_items, gr_vector_const_void_star& input_items, gr_vector_void_star&)
{
    const gr_complex* in = static_cast<const gr_complex*>(input_items[0]);
    size_t done = 0, n = (size_t)noutput_items;
    while (done < n) {
        // the scheduler may hand out any count: keep an odd sample for the next call
        d_buf.assign(d_carry.begin(), d_carry.end());
        const size_t take = std::min(n - done, kChunk - d_buf.size());
        d_buf.insert(d_buf.end(), in + done, in + done + take);
        done += take;
        const size_t even = d_buf.size() & ~(size_t)1;
        d_carry.assign(d_buf.begin() + even, d_buf.end());
        if (even) run(d_buf.data(), even);
    }
    return noutput_items;
}
std::vector<float>* gr_demod_hip::get_audio_data()
{
    gr::thread::scoped_lock g(d_mutex);      // gr_audio_sink::get_data: one packet of 640 samples, or nullptr (gr_audio_sink.cpp:51-66)
    if (d_boxa.size() < 640) return nullptr;
    auto* v = new std::vector<float>(d_boxa.begin(), d_boxa.begin() + 640);
    d_boxa.erase(d_boxa.begin(), d_boxa.begin() + 640);
    return v;
}
void gr_demod_hip::set_squelch(int value) { chk(qrl_demod_set_squelch(d_h, (double)value), "qrl_demod_set_squelch"); }
void gr_demod_hip::set_agc_attack(float value) { d_attack = value; chk(qrl_demod_set_agc(d_h, d_attack, d_decay), "qrl_demod_set_agc"); }
void gr_demod_hip::set_agc_decay(float value) { d_decay = value; chk(qrl_demod_set_agc(d_h, d_attack, d_decay), "qrl_demod_set_agc"); }
std::vector<unsigned char>* gr_demod_hip::get_data(int nr)
{
    gr::thread::scoped_lock g(d_mutex);
    std::vector<unsigned char>& box = nr == 1 ? d_box1 : d_box2;
    if (box.size() < (d_df1 ? 1u : 32u)) return nullptr;     // gr_bit_sink.cpp:48-52 (>= 32 bits); gr_deframer_bb::get_data: anything
    std::vector<unsigned char>* v = new std::vector<unsigned char>(box);
    box.clear();
    return v;
}
std::vector<gr_complex>* gr_demod_hip::get_constellation_data()
{
    gr::thread::scoped_lock g(d_mutex);
    if (d_boxc.size() < 32) return nullptr;      // gr_const_sink::get_data (src/gr/gr_const_sink.cpp:48-62)
    std::vector<gr_complex>* v = new std::vector<gr_complex>(d_boxc);
    d_boxc.clear();
    return v;
}

static gr_mod_hip_sptr make_mod(qrl_runtime& rt, int modem, int sps, int samp_rate, int carrier_freq, int filter_width, bool fm)
{
    qrl_mod_config c{};
    c.modem_type = modem; c.use_mode_defaults = 0;   // any member of the family selects the chain, the arguments configure it
    c.sps = sps; c.samp_rate = samp_rate; c.carrier_freq = carrier_freq; c.filter_width = filter_width; c.fm = fm;
    c.batch = 1; c.max_bytes = 8192; c.bb_gain = 1.0f;
    qrl_mod* h = nullptr;
    chk(qrl_mod_create(rt.ctx(), &c, &h), "qrl_mod_create");
    return gr_mod_hip_sptr(new gr_mod_hip(h));
}
gr_mod_hip_sptr make_gr_mod_qpsk_hip(qrl_runtime& rt, int sps, int samp_rate, int carrier_freq, int filter_width)
{ return make_mod(rt, QRL_MODEM_QPSK250K, sps, samp_rate, carrier_freq, filter_width, false); }
gr_mod_hip_sptr make_gr_mod_2fsk_hip(qrl_runtime& rt, int sps, int samp_rate, int carrier_freq, int filter_width, bool fm)
{ return make_mod(rt, QRL_MODEM_2FSK1K, sps, samp_rate, carrier_freq, filter_width, fm); }
gr_mod_hip_sptr make_gr_mod_gmsk_hip(qrl_runtime& rt, int sps, int samp_rate, int carrier_freq, int filter_width)
{ return make_mod(rt, QRL_MODEM_GMSK10K, sps, samp_rate, carrier_freq, filter_width, false); }
gr_mod_hip_sptr make_gr_mod_4fsk_hip(qrl_runtime& rt, int sps, int samp_rate, int carrier_freq, int filter_width, bool fm)
{ return make_mod(rt, QRL_MODEM_4FSK2KFM, sps, samp_rate, carrier_freq, filter_width, fm); }
gr_mod_hip_sptr make_gr_mod_bpsk_hip(qrl_runtime& rt, int sps, int samp_rate, int carrier_freq, int filter_width)
{ return make_mod(rt, QRL_MODEM_BPSK1K, sps, samp_rate, carrier_freq, filter_width, false); }

gr_mod_hip::gr_mod_hip(qrl_mod* handle)
    : gr::sync_interpolator("gr_mod_hip", gr::io_signature::make(1, 1, sizeof(unsigned char)),
                            gr::io_signature::make(1, 1, sizeof(gr_complex)), (unsigned)qrl_mod_samples_per_byte(handle)), d_h(handle)
{
    hchk(hipMalloc(reinterpret_cast<void**>(&d_bytes), kMaxBytes), "hipMalloc");
    hchk(hipMalloc(reinterpret_cast<void**>(&d_iq), kMaxBytes * qrl_mod_samples_per_byte(d_h) * sizeof(gr_complex)), "hipMalloc");
}
gr_mod_hip::~gr_mod_hip()
{
    if (d_h) qrl_mod_destroy(d_h);
    if (d_bytes) (void)hipFree(d_bytes);
    if (d_iq) (void)hipFree(d_iq);
}
void gr_mod_hip::set_bb_gain(float v) { chk(qrl_mod_set_bb_gain(d_h, v), "qrl_mod_set_bb_gain"); }
gr_amod_hip_sptr make_gr_mod_nbfm_hip(qrl_runtime& rt, int sps, int samp_rate, int carrier_freq, int filter_width)
{
    (void)carrier_freq;
    if (sps != 20 || samp_rate != 1000000 || (filter_width != 2500 && filter_width != 5000))
        throw std::invalid_argument("make_gr_mod_nbfm_hip: the instances of gr_mod_base.cpp:171-172 are (20, 1000000, ., 2500 | 5000)");
    return gr_amod_hip_sptr(new gr_amod_hip(rt, filter_width));
}
gr_amod_hip_sptr make_gr_mod_am_hip(qrl_runtime& rt, int sps, int samp_rate, int carrier_freq, int filter_width)
{
    (void)carrier_freq;
    if (sps != 125 || samp_rate != 1000000 || filter_width != 5000)
        throw std::invalid_argument("make_gr_mod_am_hip: the instance of gr_mod_base.cpp:167 is (125, 1000000, ., 5000)");
    return gr_amod_hip_sptr(new gr_amod_hip(rt, filter_width, QRL_MODEM_AM5000));
}
gr_amod_hip::gr_amod_hip(qrl_runtime& rt, int filter_width, int modem_type)
    : gr::sync_interpolator("gr_amod_hip", gr::io_signature::make(1, 1, sizeof(float)), gr::io_signature::make(1, 1, sizeof(gr_complex)), 125)
{
    qrl_amod_config c{};
    c.modem_type = modem_type >= 0 ? modem_type : filter_width == 2500 ? QRL_MODEM_NBFM2500 : QRL_MODEM_NBFM5000;
    c.batch = 1; c.max_samples = kMaxAudio; c.bb_gain = 1.0f;
    chk(qrl_amod_create(rt.ctx(), &c, &d_h), "qrl_amod_create");
    hchk(hipMalloc(reinterpret_cast<void**>(&d_audio), kMaxAudio * sizeof(float)), "hipMalloc");
    hchk(hipMalloc(reinterpret_cast<void**>(&d_iq), kMaxAudio * 125 * sizeof(gr_complex)), "hipMalloc");
}
gr_amod_hip::~gr_amod_hip()
{
    if (d_h) qrl_amod_destroy(d_h);
    if (d_audio) (void)hipFree(d_audio);
    if (d_iq) (void)hipFree(d_iq);
}
void gr_amod_hip::set_bb_gain(float value) { chk(qrl_amod_set_bb_gain(d_h, value), "qrl_amod_set_bb_gain"); }
// noutput_items = 125 x the audio items offered; audio that does not fill a group of 4 waits in d_carry (its 125 x output items are
// handed out with the group that completes it: the block then returns fewer items than asked, as a GNU Radio block may)
int gr_amod_hip::work(int noutput_items, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items)
{
    const float* in = static_cast<const float*>(input_items[0]);
    gr_complex* out = static_cast<gr_complex*>(output_items[0]);
    d_carry.insert(d_carry.end(), in, in + (size_t)noutput_items / 125);
    const size_t usable = d_carry.size() & ~(size_t)3;
    hipStream_t s = static_cast<hipStream_t>(qrl_amod_stream(d_h));
    size_t done = 0;
    while (done < usable) {
        const size_t take = std::min(usable - done, kMaxAudio);
        hchk(hipMemcpyAsync(d_audio, d_carry.data() + done, take * sizeof(float), hipMemcpyHostToDevice, s), "H2D");
        chk(qrl_amod_process(d_h, d_audio, kMaxAudio, take, d_iq, kMaxAudio * 125), "qrl_amod_process");
        hchk(hipMemcpyAsync(out + done * 125, d_iq, take * 125 * sizeof(gr_complex), hipMemcpyDeviceToHost, s), "D2H");
        chk(qrl_amod_sync(d_h), "qrl_amod_sync");
        done += take;
    }
    d_carry.erase(d_carry.begin(), d_carry.begin() + (std::ptrdiff_t)usable);
    return (int)(usable * 125);
}

int gr_mod_hip::work(int noutput_items, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items)
{
    const size_t spb = qrl_mod_samples_per_byte(d_h);
    const unsigned char* in = static_cast<const unsigned char*>(input_items[0]);
    gr_complex* out = static_cast<gr_complex*>(output_items[0]);
    size_t nbytes = (size_t)noutput_items / spb, done = 0;
    hipStream_t s = static_cast<hipStream_t>(qrl_mod_stream(d_h));
    while (done < nbytes) {
        const size_t take = std::min(nbytes - done, kMaxBytes);
        hchk(hipMemcpyAsync(d_bytes, in + done, take, hipMemcpyHostToDevice, s), "H2D");
        chk(qrl_mod_process(d_h, d_bytes, kMaxBytes, take, d_iq, kMaxBytes * spb), "qrl_mod_process");
        hchk(hipMemcpyAsync(out + done * spb, d_iq, take * spb * sizeof(gr_complex), hipMemcpyDeviceToHost, s), "D2H");
        chk(qrl_mod_sync(d_h), "qrl_mod_sync");
        done += take;
    }
    return (int)(nbytes * spb);
}
