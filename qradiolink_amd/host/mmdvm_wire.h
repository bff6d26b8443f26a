// mmdvm_wire.h — host side of the MMDVM multi-carrier path above the C ABI:
//   BurstTimer            TDMA slot bookkeeping                      [reference src/bursttimer.h:27-89, src/bursttimer.cpp:20-280]
//   mmdvm_sink::work      per-sample slot marks + 720-sample frames  [src/gr/gr_mmdvm_sink.cpp:66-176]
//   mmdvm_source::work    frames -> 720-sample bursts, time / zero tags [src/gr/gr_mmdvm_source.cpp:65-243]
//   zero_idle_runs        gr_zero_idle_bursts as absolute zero runs   [src/gr/gr_zero_idle_bursts.cpp:45-84]
// The DSP on either side is the HIP library (qrl_chan_* delivers the int16 channels + RSSI tags the sink consumes, qrl_synth_*
// takes the int16 bursts the source hands out and applies the zero runs on the device).  The reference's transport is ZeroMQ
// (PUSH per channel for RX, REQ/REP for TX); it is NOT part of this layer: frames leave and arrive through two std::function
// hooks, so that a maintainer binds zmq::socket_t::send / recv exactly where the reference calls them.
// Integer / byte logic only: results are byte-identical to the reference by construction of the same state machines; the Qt
// container and the wall-clock members of BurstTimer that take no part in any result (t1/t2/tx1/tx2 except the TX_TIMEOUT of
// set_tx) are not restated.
#pragma once
#include <cstddef>
#include <cstdint>
#include <deque>
#include <functional>
#include <mutex>
#include <vector>

namespace qrl_host {

constexpr int MAX_MMDVM_CHANNELS = 7;                 // bursttimer.h:25
constexpr uint64_t BURST_DELAY = 100000000ULL;        // ns, :27
constexpr uint64_t SLOT_TIME = 30000000ULL;           // :28
constexpr int32_t SAMPLES_PER_SLOT = 720;             // :30
constexpr uint64_t TIME_PER_SAMPLE = 41667ULL;        // :31
constexpr uint8_t MARK_SLOT1 = 0x08, MARK_SLOT2 = 0x04, MARK_NONE = 0x00;   // gr_mmdvm_sink.cpp:20-22
constexpr int32_t ZERO_SAMPLES = SAMPLES_PER_SLOT * 25 / 24;                // gr_mmdvm_source.cpp:23

class BurstTimer {
public:
    BurstTimer(uint64_t burst_delay = BURST_DELAY, uint64_t samples_per_slot = SAMPLES_PER_SLOT,
               uint64_t time_per_sample = TIME_PER_SAMPLE, uint64_t slot_time = SLOT_TIME);
    void set_enabled(bool v) { _enabled = v; }
    void set_params(uint64_t samples_per_slot, uint64_t time_per_sample, uint64_t slot_time, uint64_t burst_delay);
    void reset_timer(int cn = 0);
    uint64_t get_time_delta(int cn = 0);
    void set_timer(uint64_t value, int cn = 0);
    void increment_sample_counter(int cn = 0);
    uint64_t get_sample_counter(int cn);
    int check_time(int cn = 0, bool time_base_received = false);
    uint64_t allocate_slot(int slot_no, int64_t& timing, int cn = 0);
    bool get_timing_initialized(int cn);
    size_t pending_slots(int cn);

private:
    struct slot { uint8_t slot_no; uint64_t slot_time; uint64_t slot_sample_counter; };
    bool _enabled = true;
    bool _timing_initialized[MAX_MMDVM_CHANNELS] = {};
    std::mutex _timing_mutex[MAX_MMDVM_CHANNELS], _slot_mutex[MAX_MMDVM_CHANNELS];
    uint64_t _samples_per_slot, _time_per_sample, _slot_time, _burst_delay;
    uint64_t _sample_counter[MAX_MMDVM_CHANNELS] = {}, _last_slot[MAX_MMDVM_CHANNELS] = {}, _time_base[MAX_MMDVM_CHANNELS] = {};
    std::deque<slot> _slot_times[MAX_MMDVM_CHANNELS];
};

// rx_time stream tag of the SDR source: item offset inside this work() call + UHD-style (full seconds, fractional seconds)
struct time_tag { uint32_t offset; uint64_t secs; double fracs; };

// a35: gr_mmdvm_sink.  One object = all channels (like the reference block); work() is called once per scheduler pass with the
// same number of new int16 items on every channel (the block is a sync_block with cn inputs).
class mmdvm_sink {
public:
    using send_fn = std::function<void(int chan, const uint8_t* msg, size_t len)>;   // zmq PUSH "ipc:///tmp/mmdvm-rx<N>.ipc"
    mmdvm_sink(BurstTimer* burst_timer, int num_channels, bool use_tdma, send_fn send);
    // in[chan][i], i < noutput_items (<= SAMPLES_PER_SLOT: set_max_noutput_items); rssi[chan] = values of the RSSI tags
    // (rssi_tag_block, one per 300 samples) that fall into this call, in offset order; tags[chan] = rx_time tags of this call.
    int work(int noutput_items, const int16_t* const* in, const std::vector<float>* rssi, const std::vector<time_tag>* tags);

private:
    BurstTimer* _burst_timer; int _num_channels; bool _use_tdma; send_fn _send;
    std::vector<int16_t> data_buf[MAX_MMDVM_CHANNELS];
    std::vector<uint8_t> control_buf[MAX_MMDVM_CHANNELS];
    std::vector<uint32_t> _rssi[MAX_MMDVM_CHANNELS];
    uint32_t _last_rssi_on_timeslot[MAX_MMDVM_CHANNELS] = {};
    int64_t _slot_sample_counter[MAX_MMDVM_CHANNELS] = {};
};

// tags the source attaches to its output (gr_mmdvm_source.cpp:246-264)
struct tx_tag { int chan; uint32_t offset; bool is_zero; uint64_t value; };   // tx_time: value = ns; zero_samples: value = count

// a50: gr_mmdvm_source.  work() hands out exactly SAMPLES_PER_SLOT int16 items per channel.
class mmdvm_source {
public:
    // REQ/REP exchange of the reference (send "s", receive one message): returns the message length (0 = nothing), data in buf
    using request_fn = std::function<size_t(int chan, std::vector<uint8_t>& msg)>;
    mmdvm_source(BurstTimer* burst_timer, int num_channels, bool use_tdma, request_fn request);
    // out[chan][SAMPLES_PER_SLOT]; returns the number of items produced per channel (0 while the time base is not initialised in
    // TDMA mode, else SAMPLES_PER_SLOT); tags are appended.  sleep_ns receives the reference's timing-correction nanosleep.
    int work(int16_t* const* out, std::vector<tx_tag>& tags, int64_t* sleep_ns = nullptr);

private:
    void handle_idle_time(int16_t* out, int noutput_items, int which, bool add_tag, std::vector<tx_tag>& tags);
    int handle_data_bursts(int16_t* out, unsigned n, int which, bool add_tag, std::vector<tx_tag>& tags);
    BurstTimer* _burst_timer; int _num_channels; bool _use_tdma; request_fn _request;
    int _sn = 2; int64_t _timing_correction = 0;
    bool _in_tx[MAX_MMDVM_CHANNELS] = {};
    std::vector<int16_t> data_buf[MAX_MMDVM_CHANNELS];
    std::vector<uint8_t> control_buf[MAX_MMDVM_CHANNELS];
};

// gr_zero_idle_bursts (delay 0): the zero_samples tags of one channel, as absolute [start, start + count) runs at the rate of
// the block's input.  rate_num / rate_den = the relative rate between the source (24 ksps) and the block (25/24 behind the
// rational resampler of gr_mod_mmdvm_multi2, 1/1 in gr_mod_mmdvm): GNU Radio moves a tag through a rate-changing block to
// floor(offset * rate + 1/2).  A run that is still counting when the next tag arrives is restarted (the block reloads its counter).
struct zero_run { uint64_t start, count; };
std::vector<zero_run> zero_idle_runs(const std::vector<tx_tag>& tags, int chan, uint64_t items_written_24k, uint32_t rate_num, uint32_t rate_den);

}  // namespace qrl_host
