// gr_stub — TEST INFRASTRUCTURE.  The smallest stand-in for the GNU Radio runtime headers that lets the reference's OWN custom blocks
// (src/gr/gr_dmr_dmo_sink.cpp, gr_deframer_bb.cpp, gr_4fsk_discriminator.cpp, rssi_tag_block.cpp) compile unmodified, where they lie,
// into oracle/_ref/libqrl_ref.so: base classes that record what a block consumes / tags, nothing of GNU Radio's scheduler or DSP.
// Written for this repository; it contains no GNU Radio or reference code.
#pragma once
#include <algorithm>
#include <complex>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include <pmt/pmt.h>

typedef std::complex<float> gr_complex;
typedef std::vector<int> gr_vector_int;
typedef std::vector<const void*> gr_vector_const_void_star;
typedef std::vector<void*> gr_vector_void_star;

namespace gr {

namespace thread {
typedef std::mutex mutex;
typedef std::lock_guard<std::mutex> scoped_lock;
typedef std::condition_variable condition_variable;
}  // namespace thread

class io_signature {
public:
    typedef std::shared_ptr<io_signature> sptr;
    static sptr make(int, int, int) { return sptr(new io_signature); }
    static sptr makev(int, int, const std::vector<int>&) { return sptr(new io_signature); }
};

struct tag_t {
    uint64_t offset; pmt::pmt_t key, value; unsigned port = 0;
    static bool offset_compare(const tag_t& a, const tag_t& b) { return a.offset < b.offset; }
};

class block {
public:
    enum { WORK_CALLED_PRODUCE = -2, WORK_DONE = -1 };
    block() {}                                   // pure interface sub-classes (virtual inheritance), as in GNU Radio
    block(const std::string& name, io_signature::sptr, io_signature::sptr) : d_name(name) {}
    virtual ~block() {}
    virtual void forecast(int, gr_vector_int&) {}
    virtual int general_work(int, gr_vector_int&, gr_vector_const_void_star&, gr_vector_void_star&) { return 0; }
    void set_relative_rate(double) {}
    void set_alignment(int) {}
    void consume_each(int n) { stub_consumed += n; }
    void consume(int, int n) { stub_consumed += n; }
    uint64_t nitems_written(unsigned) const { return stub_written; }
    uint64_t nitems_read(unsigned) const { return stub_read; }
    void add_item_tag(unsigned port, uint64_t offset, const pmt::pmt_t& key, const pmt::pmt_t& value) { stub_tags.push_back(tag_t{offset, key, value, port}); }
    // tags of the input whose offset lies in [nitems_read + start, nitems_read + end) and whose key is `key`
    void get_tags_in_window(std::vector<tag_t>& v, unsigned, uint64_t start, uint64_t end, const pmt::pmt_t& key)
    {
        v.clear();
        for (const tag_t& t : stub_in_tags)
            if (t.offset >= stub_read + start && t.offset < stub_read + end && t.key->sym == key->sym) v.push_back(t);
    }
    // per-port form used by gr_mmdvm_sink: absolute range [lo, hi) on input `port`
    void get_tags_in_range(std::vector<tag_t>& v, unsigned port, uint64_t lo, uint64_t hi, const pmt::pmt_t& key)
    {
        v.clear();
        for (const tag_t& t : stub_in_tags)
            if (t.port == port && t.offset >= lo && t.offset < hi && t.key->sym == key->sym) v.push_back(t);
    }
    void set_min_noutput_items(int) {}
    void set_max_noutput_items(int) {}
    std::vector<tag_t> stub_in_tags;
    void set_history(unsigned) {}
    void set_output_multiple(int) {}
    void set_thread_priority(int) {}
    // what the harness reads back / advances between work() calls
    long stub_consumed = 0;
    uint64_t stub_written = 0, stub_read = 0;
    std::vector<tag_t> stub_tags;

private:
    std::string d_name;
};
class sync_block : public block {
public:
    sync_block() {}
    using block::block;
    virtual int work(int, gr_vector_const_void_star&, gr_vector_void_star&) { return 0; }
};
class sync_interpolator : public sync_block {
public:
    sync_interpolator(const std::string& name, io_signature::sptr a, io_signature::sptr b, unsigned) : sync_block(name, a, b) {}
};

}  // namespace gr

namespace boost { typedef std::mutex mutex; }   // a member the reference's gr_deframer_bb declares and never uses

namespace gnuradio {
template <class T>
std::shared_ptr<T> get_initial_sptr(T* p) { return std::shared_ptr<T>(p); }
}  // namespace gnuradio
