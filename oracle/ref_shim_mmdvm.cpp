// ref_shim_mmdvm.cpp -- TEST INFRASTRUCTURE.  The reference's OWN MMDVM wire layer -- BurstTimer (/root/reference/src/bursttimer.cpp),
// gr_mmdvm_sink::work and gr_mmdvm_source::work (src/gr/gr_mmdvm_sink.cpp, gr_mmdvm_source.cpp) -- compiled unmodified against
// oracle/gr_stub (GNU Radio base classes, pmt, an in-memory zmq.hpp, QVector) and driven through the SAME C entry points as the
// product's host layer has in tests/host/mmdvm_shim.cpp (mw_* -> ref_mw_*), so that tests/test_mmdvm_wire.py can run one scenario on
// both and compare frames, slot marks, bursts and tags byte for byte.  The build renames nanosleep (-Dnanosleep=qrl_stub_nanosleep):
// the source's timing-correction sleep is recorded instead of slept.
#include <cstdint>
#include <cstring>
#include <ctime>
#include <iostream>
#include <sstream>
#include <string>
#include <complex>
#include <memory>
#include <mutex>
#include <condition_variable>
#include <algorithm>
#include <deque>
#include <optional>
#include <chrono>
#include <vector>

#define private public      // the sockets (mailboxes here) are private members of the blocks
#include "src/gr/gr_mmdvm_sink.h"
#include "src/gr/gr_mmdvm_source.h"
#undef private

static int64_t g_slept_ns = 0;
extern "C" int qrl_stub_nanosleep(const struct timespec* req, struct timespec*) { g_slept_ns += (int64_t)req->tv_sec * 1000000000LL + req->tv_nsec; return 0; }

struct RefSink { gr_mmdvm_sink_sptr blk; int nch; uint64_t read = 0; };
struct RefSource { gr_mmdvm_source_sptr blk; int nch; uint64_t written = 0; };

extern "C" {

void* ref_mw_timer_new(void) { return new BurstTimer(); }
void ref_mw_timer_free(void* t) { delete static_cast<BurstTimer*>(t); }
void ref_mw_timer_set_params(void* t, uint64_t sps, uint64_t tps, uint64_t slot_time, uint64_t burst_delay) { static_cast<BurstTimer*>(t)->set_params(sps, tps, slot_time, burst_delay); }
void ref_mw_timer_set_timer(void* t, uint64_t ns, int cn) { static_cast<BurstTimer*>(t)->set_timer(ns, cn); }
uint64_t ref_mw_timer_allocate_slot(void* t, int slot_no, int cn, int64_t* timing) { return static_cast<BurstTimer*>(t)->allocate_slot(slot_no, *timing, cn); }
int ref_mw_timer_check_time(void* t, int cn, int time_base_received) { return static_cast<BurstTimer*>(t)->check_time(cn, time_base_received != 0); }

void* ref_mw_sink_new(void* timer, int nch, int tdma)
{
    RefSink* b = new RefSink;
    b->blk = make_gr_mmdvm_sink(static_cast<BurstTimer*>(timer), (uint8_t)nch, true, tdma != 0);
    b->nch = nch;
    return b;
}
void ref_mw_sink_free(void* s) { delete static_cast<RefSink*>(s); }
int ref_mw_sink_work(void* s, int nch, int n, const int16_t* in, const float* rssi, const int* nrssi, const uint32_t* tag_off, const uint64_t* tag_secs,
                     const double* tag_fracs, const int* ntags)
{
    RefSink* b = static_cast<RefSink*>(s);
    gr_vector_const_void_star ins(nch);
    gr_vector_void_star outs;
    b->blk->stub_in_tags.clear();
    b->blk->stub_read = b->read;
    int ro = 0, to = 0;
    for (int c = 0; c < nch; ++c) {
        ins[c] = in + (size_t)c * n;
        // RSSI tags: the wire layer only looks at their order inside the call
        for (int i = 0; i < nrssi[c]; ++i) { gr::tag_t t{b->read + (uint64_t)i, pmt::string_to_symbol("RSSI"), pmt::from_float(rssi[ro++])}; t.port = (unsigned)c; b->blk->stub_in_tags.push_back(t); }
        for (int i = 0; i < ntags[c]; ++i, ++to) {
            gr::tag_t t{b->read + tag_off[to], pmt::string_to_symbol("rx_time"), pmt::make_tuple(pmt::from_uint64(tag_secs[to]), pmt::from_double(tag_fracs[to]))};
            t.port = (unsigned)c;
            b->blk->stub_in_tags.push_back(t);
        }
    }
    const int r = b->blk->work(n, ins, outs);
    b->read += (uint64_t)n;
    return r;
}
// same layout as the product shim: {chan u32, len u32, message} per frame, in the order of sending per channel ascending
size_t ref_mw_sink_take(void* s, uint8_t* out, size_t cap, int* frames)
{
    RefSink* b = static_cast<RefSink*>(s);
    size_t at = 0; *frames = 0;
    // the product shim records frames in send order; the reference sends channel by channel inside one work() call -- the tests
    // take after every call, where both orders coincide
    for (int c = 0; c < b->nch; ++c) {
        for (auto& m : b->blk->_zmqsocket[c].sent) {
            const uint32_t cc = (uint32_t)c, l = (uint32_t)m.size();
            if (at + 8 + m.size() > cap) break;
            std::memcpy(out + at, &cc, 4); std::memcpy(out + at + 4, &l, 4); std::memcpy(out + at + 8, m.data(), m.size());
            at += 8 + m.size(); (*frames)++;
        }
        b->blk->_zmqsocket[c].sent.clear();
    }
    return at;
}

void* ref_mw_source_new(void* timer, int nch, int tdma)
{
    RefSource* b = new RefSource;
    b->blk = make_gr_mmdvm_source(static_cast<BurstTimer*>(timer), (uint8_t)nch, true, tdma != 0);
    b->nch = nch;
    return b;
}
void ref_mw_source_free(void* s) { delete static_cast<RefSource*>(s); }
void ref_mw_source_push(void* s, int chan, const uint8_t* msg, size_t len) { static_cast<RefSource*>(s)->blk->_zmqsocket[chan].inbox.emplace_back(msg, msg + len); }
int ref_mw_source_work(void* s, int nch, int16_t* out, uint64_t* tags, int cap_tags, int* ntags, int64_t* sleep_ns)
{
    RefSource* b = static_cast<RefSource*>(s);
    gr_vector_const_void_star ins;
    gr_vector_void_star outs(nch);
    for (int c = 0; c < nch; ++c) outs[c] = out + (size_t)c * SAMPLES_PER_SLOT;
    b->blk->stub_tags.clear();
    b->blk->stub_written = b->written;
    g_slept_ns = 0;
    const int r = b->blk->work(SAMPLES_PER_SLOT, ins, outs);
    for (int c = 0; c < nch; ++c) b->blk->_zmqsocket[c].sent.clear();      // the "s" requests
    if (sleep_ns) *sleep_ns = g_slept_ns;
    int n = 0;
    for (const gr::tag_t& t : b->blk->stub_tags) {
        if (n >= cap_tags) break;
        const bool zero = t.key->sym == "zero_samples";
        tags[4 * n] = t.port; tags[4 * n + 1] = t.offset - b->written; tags[4 * n + 2] = zero ? 1 : 0;
        tags[4 * n + 3] = zero ? pmt::to_uint64(t.value)
                               : pmt::to_uint64(pmt::tuple_ref(t.value, 0)) * 1000000000ULL + (uint64_t)llround(pmt::to_double(pmt::tuple_ref(t.value, 1)) * 1e9);
        ++n;
    }
    *ntags = n;
    if (r > 0) b->written += (uint64_t)r;
    return r;
}

}
