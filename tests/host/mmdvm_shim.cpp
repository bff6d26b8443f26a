// mmdvm_shim.cpp — C-callable test shim over qradiolink_amd/host/mmdvm_wire.{h,cpp} (TEST INFRASTRUCTURE: lets
// tests/test_mmdvm_wire.py drive the host classes through ctypes and compare them with the Python model tests/mmdvm_model.py).
#include <cstring>
#include <deque>
#include <vector>

#include "mmdvm_wire.h"

using namespace qrl_host;

struct SinkBox { mmdvm_sink* sink; std::vector<uint8_t> out; int frames = 0; };
struct SourceBox { mmdvm_source* src; std::deque<std::vector<uint8_t>> queue[MAX_MMDVM_CHANNELS]; };

extern "C" {

void* mw_timer_new(void) { return new BurstTimer(); }
void mw_timer_free(void* t) { delete static_cast<BurstTimer*>(t); }
void mw_timer_set_params(void* t, uint64_t sps, uint64_t tps, uint64_t slot_time, uint64_t burst_delay) { static_cast<BurstTimer*>(t)->set_params(sps, tps, slot_time, burst_delay); }
void mw_timer_set_timer(void* t, uint64_t ns, int cn) { static_cast<BurstTimer*>(t)->set_timer(ns, cn); }
uint64_t mw_timer_allocate_slot(void* t, int slot_no, int cn, int64_t* timing) { return static_cast<BurstTimer*>(t)->allocate_slot(slot_no, *timing, cn); }
int mw_timer_check_time(void* t, int cn, int time_base_received) { return static_cast<BurstTimer*>(t)->check_time(cn, time_base_received != 0); }
uint64_t mw_timer_pending(void* t, int cn) { return static_cast<BurstTimer*>(t)->pending_slots(cn); }

void* mw_sink_new(void* timer, int nch, int tdma)
{
    SinkBox* b = new SinkBox;
    b->sink = new mmdvm_sink(static_cast<BurstTimer*>(timer), nch, tdma != 0, [b](int chan, const uint8_t* msg, size_t len) {
        const uint32_t c = (uint32_t)chan, l = (uint32_t)len;
        const size_t at = b->out.size();
        b->out.resize(at + 8 + len);
        std::memcpy(b->out.data() + at, &c, 4); std::memcpy(b->out.data() + at + 4, &l, 4); std::memcpy(b->out.data() + at + 8, msg, len);
        b->frames++;
    });
    return b;
}
void mw_sink_free(void* s) { SinkBox* b = static_cast<SinkBox*>(s); delete b->sink; delete b; }
// in [nch][n]; rssi concatenated per channel with counts nrssi[nch]; tags concatenated per channel with counts ntags[nch]
int mw_sink_work(void* s, int nch, int n, const int16_t* in, const float* rssi, const int* nrssi, const uint32_t* tag_off, const uint64_t* tag_secs,
                 const double* tag_fracs, const int* ntags)
{
    SinkBox* b = static_cast<SinkBox*>(s);
    std::vector<const int16_t*> ptr(nch);
    std::vector<std::vector<float>> rs(nch);
    std::vector<std::vector<time_tag>> tg(nch);
    int ro = 0, to = 0;
    for (int c = 0; c < nch; ++c) {
        ptr[c] = in + (size_t)c * n;
        for (int i = 0; i < nrssi[c]; ++i) rs[c].push_back(rssi[ro++]);
        for (int i = 0; i < ntags[c]; ++i, ++to) tg[c].push_back({tag_off[to], tag_secs[to], tag_fracs[to]});
    }
    return b->sink->work(n, ptr.data(), rs.data(), tg.data());
}
size_t mw_sink_take(void* s, uint8_t* out, size_t cap, int* frames)
{
    SinkBox* b = static_cast<SinkBox*>(s);
    const size_t n = b->out.size() < cap ? b->out.size() : cap;
    std::memcpy(out, b->out.data(), n);
    *frames = b->frames;
    b->out.clear(); b->frames = 0;
    return n;
}

void* mw_source_new(void* timer, int nch, int tdma)
{
    SourceBox* b = new SourceBox;
    b->src = new mmdvm_source(static_cast<BurstTimer*>(timer), nch, tdma != 0, [b](int chan, std::vector<uint8_t>& msg) -> size_t {
        if (b->queue[chan].empty()) return 0;
        msg = b->queue[chan].front();
        b->queue[chan].pop_front();
        return msg.size();
    });
    return b;
}
void mw_source_free(void* s) { SourceBox* b = static_cast<SourceBox*>(s); delete b->src; delete b; }
void mw_source_push(void* s, int chan, const uint8_t* msg, size_t len) { static_cast<SourceBox*>(s)->queue[chan].emplace_back(msg, msg + len); }
// out [nch][720]; tags: rows of {chan, offset, is_zero, value} as uint64; returns items per channel, *ntags rows written
int mw_source_work(void* s, int nch, int16_t* out, uint64_t* tags, int cap_tags, int* ntags, int64_t* sleep_ns)
{
    SourceBox* b = static_cast<SourceBox*>(s);
    std::vector<int16_t*> ptr(nch);
    for (int c = 0; c < nch; ++c) ptr[c] = out + (size_t)c * SAMPLES_PER_SLOT;
    std::vector<tx_tag> t;
    const int r = b->src->work(ptr.data(), t, sleep_ns);
    *ntags = (int)t.size() < cap_tags ? (int)t.size() : cap_tags;
    for (int i = 0; i < *ntags; ++i) { tags[4 * i] = (uint64_t)t[i].chan; tags[4 * i + 1] = t[i].offset; tags[4 * i + 2] = t[i].is_zero; tags[4 * i + 3] = t[i].value; }
    return r;
}
// zero_idle_runs over a tag table as produced above; runs out as {start, count}
int mw_zero_runs(const uint64_t* tags, int ntags, int chan, uint64_t items_written, uint32_t num, uint32_t den, uint64_t* runs, int cap)
{
    std::vector<tx_tag> t;
    for (int i = 0; i < ntags; ++i) t.push_back({(int)tags[4 * i], (uint32_t)tags[4 * i + 1], tags[4 * i + 2] != 0, tags[4 * i + 3]});
    const auto r = zero_idle_runs(t, chan, items_written, num, den);
    int n = 0;
    for (const auto& z : r) { if (n >= cap) break; runs[2 * n] = z.start; runs[2 * n + 1] = z.count; ++n; }
    return n;
}

}  // extern "C"
