// mmdvm_wire.cpp — see mmdvm_wire.h.  Every function cites the reference lines it restates.
#include "mmdvm_wire.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace qrl_host {

// ---------------------------------------------------------------------------------------------- BurstTimer (src/bursttimer.cpp)
BurstTimer::BurstTimer(uint64_t burst_delay, uint64_t samples_per_slot, uint64_t time_per_sample, uint64_t slot_time)
    : _samples_per_slot(samples_per_slot), _time_per_sample(time_per_sample), _slot_time(slot_time),
      _burst_delay(burst_delay * 1000000ULL)   // :27 -- the constructor scales its argument (set_params does not, :171-178)
{
}
void BurstTimer::set_params(uint64_t samples_per_slot, uint64_t time_per_sample, uint64_t slot_time, uint64_t burst_delay)
{
    _samples_per_slot = samples_per_slot; _time_per_sample = time_per_sample; _slot_time = slot_time; _burst_delay = burst_delay;
}
uint64_t BurstTimer::get_time_delta(int cn)   // :180-187
{
    std::lock_guard<std::mutex> g(_timing_mutex[cn]);
    return _time_base[cn] + _sample_counter[cn] * _time_per_sample;
}
void BurstTimer::reset_timer(int cn)   // :189-196
{
    std::lock_guard<std::mutex> g(_timing_mutex[cn]);
    _sample_counter[cn] = 0; _time_base[cn] = 0;
}
void BurstTimer::set_timer(uint64_t value, int cn)   // :198-206
{
    std::lock_guard<std::mutex> g(_timing_mutex[cn]);
    _sample_counter[cn] = 0; _time_base[cn] = value; _timing_initialized[cn] = true;
}
bool BurstTimer::get_timing_initialized(int cn)   // :208-212 (the reference takes mutex 0 whatever cn is)
{
    std::lock_guard<std::mutex> g(_timing_mutex[0]);
    return _timing_initialized[cn];
}
void BurstTimer::increment_sample_counter(int cn)   // :214-219
{
    std::lock_guard<std::mutex> g(_timing_mutex[cn]);
    _sample_counter[cn]++;
}
uint64_t BurstTimer::get_sample_counter(int cn)   // :221-225
{
    std::lock_guard<std::mutex> g(_timing_mutex[cn]);
    return _time_base[cn] + _sample_counter[cn] * _time_per_sample;
}
size_t BurstTimer::pending_slots(int cn)
{
    std::lock_guard<std::mutex> g(_slot_mutex[cn]);
    return _slot_times[cn].size();
}
int BurstTimer::check_time(int cn, bool time_base_received)   // :228-262
{
    if (!_enabled) return 0;
    std::lock_guard<std::mutex> g(_slot_mutex[cn]);
    if (_slot_times[cn].empty()) return 0;
    slot& s = _slot_times[cn].front();
    std::lock_guard<std::mutex> gt(_timing_mutex[cn]);
    if (!time_base_received) _sample_counter[cn]++;
    const uint64_t sample_time = _time_base[cn] + _sample_counter[cn] * _time_per_sample;
    if (sample_time >= s.slot_time && s.slot_sample_counter == 0) {
        s.slot_sample_counter++;
        return s.slot_no;
    } else if (sample_time >= s.slot_time) {
        if (s.slot_sample_counter >= (_samples_per_slot - 1)) {
            _slot_times[cn].pop_front();
            return 0;
        }
        s.slot_sample_counter++;
    }
    return 0;
}
uint64_t BurstTimer::allocate_slot(int slot_no, int64_t& timing, int cn)   // :264-299
{
    if (!_enabled) return 0;
    slot s;
    s.slot_no = (uint8_t)slot_no;
    const uint64_t elapsed = get_time_delta(0);   // (channel 0's clock for every channel, as in the reference)
    if (elapsed <= _last_slot[cn]) {
        if (cn == 0) timing = (int64_t)(_last_slot[cn] - elapsed);
        _last_slot[cn] = _last_slot[cn] + _slot_time;
    } else if (_last_slot[cn] == 0) {
        _last_slot[cn] = elapsed;
    } else if ((elapsed - _last_slot[cn]) >= _slot_time) {
        _last_slot[cn] = elapsed;
    } else {
        _last_slot[cn] = _last_slot[cn] + _slot_time;
    }
    const uint64_t nsec = _last_slot[cn] + _burst_delay;
    s.slot_time = nsec;
    s.slot_sample_counter = 0;
    std::lock_guard<std::mutex> g(_slot_mutex[cn]);
    _slot_times[cn].push_back(s);
    return nsec;
}

// ---------------------------------------------------------------------------------------------- gr_mmdvm_sink (a35)
mmdvm_sink::mmdvm_sink(BurstTimer* burst_timer, int num_channels, bool use_tdma, send_fn send)
    : _burst_timer(burst_timer), _num_channels(num_channels), _use_tdma(use_tdma), _send(std::move(send))
{
    for (int i = 0; i < _num_channels; ++i) {   // gr_mmdvm_sink.cpp:42-56
        data_buf[i].reserve(2 * SAMPLES_PER_SLOT);
        control_buf[i].reserve(2 * SAMPLES_PER_SLOT);
        _rssi[i].reserve(SAMPLES_PER_SLOT);
    }
}

int mmdvm_sink::work(int noutput_items, const int16_t* const* in, const std::vector<float>* rssi, const std::vector<time_tag>* tags)
{
    for (int chan = 0; chan < _num_channels; ++chan) {   // :77
        std::vector<time_tag> t = tags ? tags[chan] : std::vector<time_tag>();
        std::sort(t.begin(), t.end(), [](const time_tag& a, const time_tag& b) { return a.offset < b.offset; });   // :83-87
        if (rssi) for (float v : rssi[chan]) _rssi[chan].push_back((uint32_t)std::fabs(v));                       // :93-97
        for (int i = 0; i < noutput_items; ++i) {   // :99
            bool time_base_received = false;
            if (_slot_sample_counter[chan] > 0) _slot_sample_counter[chan]++;
            for (const time_tag& tag : t) {
                if (tag.offset == (uint32_t)i) {   // :106-116
                    const uint64_t time = (uint64_t)std::llround((double)(tag.secs * 1000000000ULL) + (tag.fracs * 1000000000.0));
                    _burst_timer->set_timer(time, chan);
                    time_base_received = true;
                    break;
                }
            }
            uint8_t control = MARK_NONE;
            const int slot_no = _burst_timer->check_time(chan, time_base_received);   // :121
            if (slot_no == 1) { control = MARK_SLOT1; _slot_sample_counter[chan] = 1; }
            if (slot_no == 2) { control = MARK_SLOT2; _slot_sample_counter[chan] = 1; }
            control_buf[chan].push_back(control);
            data_buf[chan].push_back(in[chan][i]);
            if (_slot_sample_counter[chan] >= SAMPLES_PER_SLOT) {   // :137-150: the lower of the last two RSSI tags of the slot
                // (the reference calls back() on a vector that rssi_tag_block keeps non-empty: one tag per 300 samples; an empty
                //  one is undefined behaviour there and reads as 0 here)
                const uint32_t rssi1 = _rssi[chan].empty() ? 0u : _rssi[chan].back();
                uint32_t rssi2 = 32767;
                if (_rssi[chan].size() > 1) { _rssi[chan].pop_back(); rssi2 = _rssi[chan].back(); }
                _last_rssi_on_timeslot[chan] = rssi1 < rssi2 ? rssi1 : rssi2;
                _rssi[chan].clear();
                _slot_sample_counter[chan] = 0;
            }
        }
        // buffer up to two timeslots before sending samples to MMDVM (:152-172): {u32 n, u32 rssi, u8 control[n], i16 data[n]}
        if (data_buf[chan].size() >= (size_t)SAMPLES_PER_SLOT) {
            const uint32_t num_items = SAMPLES_PER_SLOT;
            std::vector<uint8_t> msg(2 * sizeof(uint32_t) + num_items * sizeof(uint8_t) + num_items * sizeof(int16_t));
            std::memcpy(msg.data(), &num_items, sizeof(uint32_t));
            std::memcpy(msg.data() + sizeof(uint32_t), &_last_rssi_on_timeslot[chan], sizeof(uint32_t));
            std::memcpy(msg.data() + 2 * sizeof(uint32_t), control_buf[chan].data(), num_items);
            std::memcpy(msg.data() + 2 * sizeof(uint32_t) + num_items, data_buf[chan].data(), num_items * sizeof(int16_t));
            if (_send) _send(chan, msg.data(), msg.size());
            data_buf[chan].erase(data_buf[chan].begin(), data_buf[chan].begin() + num_items);
            control_buf[chan].erase(control_buf[chan].begin(), control_buf[chan].begin() + num_items);
            _last_rssi_on_timeslot[chan] = 0;
        }
    }
    return noutput_items;
}

// ---------------------------------------------------------------------------------------------- gr_mmdvm_source (a50)
mmdvm_source::mmdvm_source(BurstTimer* burst_timer, int num_channels, bool use_tdma, request_fn request)
    : _burst_timer(burst_timer), _num_channels(num_channels), _use_tdma(use_tdma), _request(std::move(request))
{
}

void mmdvm_source::handle_idle_time(int16_t* out, int noutput_items, int which, bool add_tag, std::vector<tx_tag>& tags)   // :112-128
{
    _sn = _sn == 2 ? 1 : 2;                                          // alternate_slots, :172-178
    tags.push_back({which, 0u, true, (uint64_t)ZERO_SAMPLES});        // add_zero_tag(0, ZERO_SAMPLES, which)
    for (int i = 0; i < noutput_items; ++i) {
        out[i] = 0;
        if (i == 710) {
            const uint64_t time = _burst_timer->allocate_slot(_sn, _timing_correction, which);
            if (time > 0 && add_tag) tags.push_back({which, (uint32_t)i, false, time});
        }
    }
}

int mmdvm_source::handle_data_bursts(int16_t* out, unsigned n, int which, bool add_tag, std::vector<tx_tag>& tags)   // :130-170
{
    int num_tags_added = 0;
    for (unsigned i = 0; i < n; ++i)
        if (control_buf[which][i] == MARK_SLOT1 || control_buf[which][i] == MARK_SLOT2) num_tags_added++;
    for (unsigned i = 0; i < n; ++i) {
        const uint8_t control = control_buf[which][i];
        out[i] = data_buf[which][i];
        if (control == MARK_SLOT1) {
            _sn = 1;
            const uint64_t time = _burst_timer->allocate_slot(1, _timing_correction, which);
            if (time > 0 && add_tag) tags.push_back({which, i, false, time});
        }
        if (control == MARK_SLOT2) {
            _sn = 2;
            const uint64_t time = _burst_timer->allocate_slot(2, _timing_correction, which);
            if (time > 0 && add_tag) tags.push_back({which, i, false, time});
        }
    }
    return num_tags_added;
}

int mmdvm_source::work(int16_t* const* out, std::vector<tx_tag>& tags, int64_t* sleep_ns)   // :180-243
{
    const int noutput_items = SAMPLES_PER_SLOT;   // set_min/max_noutput_items(SAMPLES_PER_SLOT), :56-57
    if (sleep_ns) *sleep_ns = 0;
    bool start = true;
    for (int i = 0; i < _num_channels; ++i) {
        if (!_burst_timer->get_timing_initialized(i)) {   // "Waiting for RX samples to initialize timebase"
            control_buf[i].clear();
            data_buf[i].clear();
            start = false;
        }
    }
    if (!start && _use_tdma) return 0;
    if (!start) {
        for (int i = 0; i < _num_channels; ++i) std::fill(out[i], out[i] + noutput_items, (int16_t)0);
        return SAMPLES_PER_SLOT;
    }
    for (int j = 0; j < _num_channels; ++j) {   // get_zmq_message, :65-110: {u32 n, u8 control[n], i16 data[n]}
        std::vector<uint8_t> msg;
        const size_t size = _request ? _request(j, msg) : 0;
        if (size < 1) { _in_tx[j] = false; continue; }
        uint32_t buf_size = 0;
        if (size >= sizeof(uint32_t)) std::memcpy(&buf_size, msg.data(), sizeof(uint32_t));
        if (buf_size > 0 && size >= sizeof(uint32_t) + (size_t)buf_size * 3) {
            _in_tx[j] = true;
            const uint8_t* control = msg.data() + sizeof(uint32_t);
            const uint8_t* data = control + buf_size;
            for (uint32_t i = 0; i < buf_size; ++i) {
                int16_t v;
                std::memcpy(&v, data + 2 * (size_t)i, sizeof v);
                control_buf[j].push_back(control[i]);
                data_buf[j].push_back(v);
            }
        } else {
            _in_tx[j] = false;
        }
    }
    if (_timing_correction > 0) {   // the reference nanosleeps here; the caller decides (:203-208)
        if (sleep_ns) *sleep_ns = _timing_correction;
        _timing_correction = 0;
    }
    for (int i = 0; i < _num_channels; ++i)
        if (data_buf[i].empty()) handle_idle_time(out[i], noutput_items, i, i == 0, tags);
    for (int i = 0; i < _num_channels; ++i) {
        const unsigned n = (unsigned)std::min<size_t>(data_buf[i].size(), (size_t)noutput_items);
        handle_data_bursts(out[i], n, i, i == 0, tags);
        data_buf[i].erase(data_buf[i].begin(), data_buf[i].begin() + n);
        control_buf[i].erase(control_buf[i].begin(), control_buf[i].begin() + n);
    }
    return SAMPLES_PER_SLOT;
}

// ---------------------------------------------------------------------------------------------- gr_zero_idle_bursts
std::vector<zero_run> zero_idle_runs(const std::vector<tx_tag>& tags, int chan, uint64_t items_written_24k, uint32_t rate_num, uint32_t rate_den)
{
    std::vector<zero_run> runs;
    for (const tx_tag& t : tags) {
        if (!t.is_zero || t.chan != chan) continue;
        const uint64_t abs24 = items_written_24k + t.offset;
        const uint64_t start = (2 * abs24 * rate_num + rate_den) / (2ULL * rate_den);   // floor(x * num / den + 1/2)
        if (!runs.empty() && start < runs.back().start + runs.back().count) {
            // the counter is reloaded while it is still running (gr_zero_idle_bursts.cpp:62-70): the run now ends at start + count
            runs.back().count = start + t.value - runs.back().start;
        } else {
            runs.push_back({start, t.value});
        }
    }
    return runs;
}

}  // namespace qrl_host
