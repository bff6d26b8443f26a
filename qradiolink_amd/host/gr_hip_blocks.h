// gr_hip_blocks.h — C++ host side above the C ABI: GNU Radio-shaped blocks that a QRadioLink maintainer drops
// where the per-mode hier blocks sit today.  Same factory arguments as the reference factories
// (make_gr_demod_2fsk / make_gr_demod_gmsk / make_gr_demod_qpsk / make_gr_mod_qpsk), same work() block ABI
// (src/gr/gr_4fsk_discriminator.h:19-21), same mailbox ownership rules as gr_bit_sink::get_data
// (src/gr/gr_bit_sink.cpp:45-59: heap vector the caller deletes, nullptr = nothing yet) and the same error
// behaviour as device construction in the reference (std::runtime_error, src/radiocontroller.cpp:1974-1983).
#pragma once
#include "gr_compat.h"
#include "qrl_hip.h"
#include <memory>
#include <stdexcept>
#include <vector>

class qrl_runtime {   // process-wide qrl_init / qrl_shutdown (one GNU Radio "top_block" worth of device context)
public:
    explicit qrl_runtime(int device = 0);
    ~qrl_runtime();
    qrl_ctx* ctx() const { return d_ctx; }
private:
    qrl_ctx* d_ctx = nullptr;
};

class gr_demod_hip;
typedef std::shared_ptr<gr_demod_hip> gr_demod_hip_sptr;
// replaces make_gr_demod_2fsk(sps, samp_rate, carrier_freq, filter_width, fm)   src/gr/gr_demod_2fsk.cpp:19-26
gr_demod_hip_sptr make_gr_demod_2fsk_hip(qrl_runtime& rt, int sps = 125, int samp_rate = 250000, int carrier_freq = 1700,
                                         int filter_width = 8000, bool fm = false);
// replaces make_gr_demod_gmsk(sps, samp_rate, carrier_freq, filter_width)        src/gr/gr_demod_gmsk.cpp:19-26
gr_demod_hip_sptr make_gr_demod_gmsk_hip(qrl_runtime& rt, int sps = 125, int samp_rate = 250000, int carrier_freq = 1700,
                                         int filter_width = 8000);
// replaces make_gr_demod_qpsk(sps, samp_rate, carrier_freq, filter_width)        src/gr/gr_demod_qpsk.cpp:20-27
gr_demod_hip_sptr make_gr_demod_qpsk_hip(qrl_runtime& rt, int sps = 125, int samp_rate = 250000, int carrier_freq = 1700,
                                         int filter_width = 8000);

// replaces make_gr_demod_4fsk(sps, samp_rate, carrier_freq, filter_width, fm)   src/gr/gr_demod_4fsk.cpp:19-27 (FM variants)
gr_demod_hip_sptr make_gr_demod_4fsk_hip(qrl_runtime& rt, int sps = 125, int samp_rate = 250000, int carrier_freq = 1700,
                                         int filter_width = 8000, bool fm = true);
// replaces make_gr_demod_bpsk(sps, samp_rate, carrier_freq, filter_width)        src/gr/gr_demod_bpsk.cpp:19-27
gr_demod_hip_sptr make_gr_demod_bpsk_hip(qrl_runtime& rt, int sps = 125, int samp_rate = 250000, int carrier_freq = 1700,
                                         int filter_width = 8000);
// replaces make_gr_demod_dmr(sps, samp_rate)                                      src/gr/gr_demod_dmr.cpp:19-27 (port 2 = dibits)
gr_demod_hip_sptr make_gr_demod_dmr_hip(qrl_runtime& rt, int sps = 5, int samp_rate = 1000000);
// replaces make_gr_demod_m17(sps, samp_rate, carrier_freq, filter_width)            src/gr/gr_demod_m17.cpp:19-27, defaults gr_demod_m17.h:41-42 (port 2 = dibits)
gr_demod_hip_sptr make_gr_demod_m17_hip(qrl_runtime& rt, int sps = 125, int samp_rate = 1000000, int carrier_freq = 1700, int filter_width = 9000);
// replaces make_gr_demod_dsss(sps, samp_rate, carrier_freq, filter_width)           src/gr/gr_demod_dsss.cpp:21-28, instance gr_demod_base.cpp:218 (25, 1000000, 1700, 150)
gr_demod_hip_sptr make_gr_demod_dsss_hip(qrl_runtime& rt, int sps = 25, int samp_rate = 1000000, int carrier_freq = 1700, int filter_width = 150);

// replace make_gr_demod_nbfm / make_gr_demod_am / make_gr_demod_wbfm(signature, sps, samp_rate, carrier_freq, filter_width)
// src/gr/gr_demod_nbfm.cpp:19-27, gr_demod_am.cpp:19-27, gr_demod_wbfm.cpp:19-27 (port 1 = audio at 8 ksps -> get_audio_data())
gr_demod_hip_sptr make_gr_demod_nbfm_hip(qrl_runtime& rt, int sps = 125, int samp_rate = 1000000, int carrier_freq = 1700, int filter_width = 5000);
gr_demod_hip_sptr make_gr_demod_am_hip(qrl_runtime& rt, int sps = 125, int samp_rate = 1000000, int carrier_freq = 1700, int filter_width = 5000);
gr_demod_hip_sptr make_gr_demod_wbfm_hip(qrl_runtime& rt, int sps = 125, int samp_rate = 1000000, int carrier_freq = 1700, int filter_width = 75000);

// replaces make_gr_demod_ssb(signature, sps, samp_rate, carrier_freq, filter_width, sb)   src/gr/gr_demod_ssb.cpp:19-27 (sb 0 = USB, 1 = LSB)
gr_demod_hip_sptr make_gr_demod_ssb_hip(qrl_runtime& rt, int sps = 125, int samp_rate = 1000000, int carrier_freq = 1700, int filter_width = 2700, int sb = 0);

class gr_demod_hip : public gr::sync_block {
public:
    gr_demod_hip(qrl_runtime& rt, int modem_family, int sps, int samp_rate, int carrier_freq, int filter_width, bool fm);
    ~gr_demod_hip() override;
    // gr_demod_base::set_samp_rate / set_carrier_offset (src/gr/gr_demod_base.cpp:1303-1362, 1220-1225)
    void set_device_samp_rate(int device_samp_rate);
    void set_carrier_offset(double hz);
    // one cf32 input stream at the DEVICE rate (what feeds _rotator in the reference), no stream outputs: the
    // ports of the hier block are mailboxes, as they end in sinks in the reference (gr_bit_sink, gr_const_sink)
    int work(int noutput_items, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items) override;
    std::vector<unsigned char>* get_data(int nr);          // port 2 (nr = 1) / port 3 (nr = 2); caller deletes
    std::vector<gr_complex>* get_constellation_data();     // port 1; caller deletes
    std::vector<float>* get_audio_data();                  // analogue modes, port 1 (gr_audio_sink::get_data, src/gr/gr_audio_sink.cpp); caller deletes
    void set_squelch(int value);                           // gr_demod_nbfm/am/wbfm::set_squelch
    void set_agc_attack(float value);                      // gr_demod_am::set_agc_attack / set_agc_decay
    void set_agc_decay(float value);
    void flush();                                          // qrl_demod_reset + drop mailboxes
    // gr_demod_base connects ports 2/3 of the 1k/2k/10k modes to gr_deframer_bb(2 | 1 | 3) (src/gr/gr_demod_base.cpp:171-178,
    // 577-603): with a deframer attached get_data(nr) hands out what gr_deframer_bb::get_data would (sync bits + frame bits)
    void attach_deframer(int deframer_type);
private:
    void open();
    void run(const gr_complex* x, size_t n);
    qrl_runtime& d_rt;
    qrl_demod_config d_cfg{};
    qrl_demod* d_h = nullptr;
    qrl_deframer *d_df1 = nullptr, *d_df2 = nullptr; uint8_t *d_fa = nullptr, *d_fb = nullptr; uint32_t* d_fcnt = nullptr; size_t d_dfcap = 0;
    int d_df_type = 0;
    size_t d_fcap = 0, d_ccap = 0, d_bcap = 0;
    std::vector<gr_complex> d_carry;                        // at most one sample: the ABI takes even counts
    std::vector<gr_complex> d_buf;
    float *d_iq = nullptr, *d_const = nullptr; uint8_t *d_a = nullptr, *d_b = nullptr; uint32_t* d_cnt = nullptr;   // device
    void* d_cs = nullptr; uint32_t* d_hcnt = nullptr;       // copy stream (hipStream_t) the deframers run on behind the demodulator; pinned [6]: port counts + deframed counts
    std::vector<unsigned char> d_box1, d_box2, d_ha, d_hb; std::vector<gr_complex> d_boxc, d_hc;
    float* d_audio = nullptr; size_t d_acap = 0; std::vector<float> d_boxa, d_hau; float d_attack = 0.1f, d_decay = 0.1f;
    gr::thread::mutex d_mutex;
    static constexpr size_t kChunk = 1 << 18;
};

class gr_mod_hip;
typedef std::shared_ptr<gr_mod_hip> gr_mod_hip_sptr;
// replaces make_gr_mod_qpsk(sps, samp_rate, carrier_freq, filter_width)          src/gr/gr_mod_qpsk.cpp:19-30
gr_mod_hip_sptr make_gr_mod_qpsk_hip(qrl_runtime& rt, int sps = 125, int samp_rate = 250000, int carrier_freq = 1700,
                                     int filter_width = 8000);
// replaces make_gr_mod_2fsk / make_gr_mod_gmsk / make_gr_mod_4fsk / make_gr_mod_bpsk (src/gr/gr_mod_2fsk.cpp:19-28,
// gr_mod_gmsk.cpp:19-28, gr_mod_4fsk.cpp:19-28, gr_mod_bpsk.cpp:19-27); same argument lists
gr_mod_hip_sptr make_gr_mod_2fsk_hip(qrl_runtime& rt, int sps = 125, int samp_rate = 250000, int carrier_freq = 1700,
                                     int filter_width = 8000, bool fm = false);
gr_mod_hip_sptr make_gr_mod_gmsk_hip(qrl_runtime& rt, int sps = 125, int samp_rate = 250000, int carrier_freq = 1700,
                                     int filter_width = 8000);
gr_mod_hip_sptr make_gr_mod_4fsk_hip(qrl_runtime& rt, int sps = 125, int samp_rate = 250000, int carrier_freq = 1700,
                                     int filter_width = 8000, bool fm = true);
gr_mod_hip_sptr make_gr_mod_bpsk_hip(qrl_runtime& rt, int sps = 125, int samp_rate = 250000, int carrier_freq = 1700,
                                     int filter_width = 8000);
// replaces make_gr_mod_nbfm(sps, samp_rate, carrier_freq, filter_width)           src/gr/gr_mod_nbfm.cpp:19-25 (f32 audio at 8 ksps in, cf32 at
// 1 Msps out: 125 output items per input item; the scheduler hands work() whole groups, as for any sync_interpolator)
class gr_amod_hip;
typedef std::shared_ptr<gr_amod_hip> gr_amod_hip_sptr;
gr_amod_hip_sptr make_gr_mod_nbfm_hip(qrl_runtime& rt, int sps = 20, int samp_rate = 1000000, int carrier_freq = 1700, int filter_width = 5000);
// replaces make_gr_mod_am(sps, samp_rate, carrier_freq, filter_width)             src/gr/gr_mod_am.cpp:19-24, instance gr_mod_base.cpp:167 (125, 1000000, 1700, 5000)
gr_amod_hip_sptr make_gr_mod_am_hip(qrl_runtime& rt, int sps = 125, int samp_rate = 1000000, int carrier_freq = 1700, int filter_width = 5000);
class gr_amod_hip : public gr::sync_interpolator {
public:
    gr_amod_hip(qrl_runtime& rt, int filter_width, int modem_type = -1);   // modem_type -1: NBFM by filter width
    ~gr_amod_hip() override;
    void set_bb_gain(float value);                  // gr_mod_nbfm::set_bb_gain
    int work(int noutput_items, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items) override;
private:
    qrl_amod* d_h = nullptr; float *d_audio = nullptr, *d_iq = nullptr;
    std::vector<float> d_carry;                     // the C ABI takes multiples of 4 audio samples (25:4 resampler)
    static constexpr size_t kMaxAudio = 4096;
};

class gr_mod_hip : public gr::sync_interpolator {   // u8 packed bytes in -> cf32 out, qrl_mod_samples_per_byte samples per byte
public:
    gr_mod_hip(qrl_mod* handle);                    // takes ownership of a handle made by one of the factories
    ~gr_mod_hip() override;
    void set_bb_gain(float value);                  // gr_mod_qpsk::set_bb_gain
    int work(int noutput_items, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items) override;
private:
    qrl_mod* d_h = nullptr; uint8_t* d_bytes = nullptr; float* d_iq = nullptr;
    static constexpr size_t kMaxBytes = 8192;
};
