// rec_stub — TEST INFRASTRUCTURE.  Recording stand-ins for the stock GNU Radio blocks the reference's hier blocks (src/gr/gr_demod_*.cpp,
// gr_mod_*.cpp) instantiate: every ::make(...) call, every firdes design, every connect() is written to a log; nothing is computed.
// Running the reference's REAL constructors against these headers yields each chain's blocks, arguments and wiring as the reference
// has them (oracle/ref_shim_rec.cpp, tests/test_ref_chains.py).  Written for this repository; no GNU Radio or reference code.
#pragma once
#include <complex>
#include <cstdint>
#include <cstdio>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <type_traits>
#include <vector>
#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <iostream>
#include <mutex>

#include <pmt/pmt.h>          // oracle/gr_stub (second on the include path)

typedef std::complex<float> gr_complex;
typedef std::vector<int> gr_vector_int;
typedef std::vector<const void*> gr_vector_const_void_star;
typedef std::vector<void*> gr_vector_void_star;
namespace boost { typedef std::mutex mutex; }
namespace gr { namespace thread {
typedef std::mutex mutex;
typedef std::lock_guard<std::mutex> scoped_lock;
typedef std::condition_variable condition_variable;
} }

namespace gr {
namespace rec {

struct State {
    std::vector<std::string> lines;                       // "#id type(args)" / "#a:p -> #b:q" / "#id.method(args)"
    std::map<uint64_t, std::string> designs;              // hash of a tap vector -> the firdes call that made it
    int next = 1;
};
inline State& st() { static State s; return s; }

inline uint64_t hash_bytes(const void* p, size_t n)
{
    uint64_t h = 1469598103934665603ull;
    const unsigned char* b = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
    return h ^ n;
}
inline std::string num(double v) { char b[64]; std::snprintf(b, sizeof b, "%.9g", v); return b; }

struct object { int id = 0; virtual ~object() {} };

template <class T> struct is_shared : std::false_type {};
template <class T> struct is_shared<std::shared_ptr<T>> : std::true_type {};

inline std::string fmt(bool v) { return v ? "true" : "false"; }
inline std::string fmt(const char* s) { return std::string("\"") + s + "\""; }
inline std::string fmt(const std::string& s) { return "\"" + s + "\""; }
inline std::string fmt(const gr_complex& c) { return "(" + num(c.real()) + "," + num(c.imag()) + ")"; }
// floats as %.9g, doubles as %.17g (both round-trip: equal text <=> equal value), integers as integers
inline std::string fmt(float v) { char b[64]; std::snprintf(b, sizeof b, "%.9g", (double)v); return b; }
inline std::string fmt(double v) { char b[64]; std::snprintf(b, sizeof b, "%.17g", v); return b; }
inline std::string fmt(long double v) { return fmt((double)v); }
template <class T, typename std::enable_if<std::is_integral<T>::value && !std::is_same<T, bool>::value, int>::type = 0>
std::string fmt(T v) { return std::to_string((long long)v); }
template <class T, typename std::enable_if<std::is_enum<T>::value, int>::type = 0>
std::string fmt(T v) { return "enum:" + std::to_string((long long)v); }
template <class T> std::string fmt(const std::shared_ptr<T>& p) { return p ? "#" + std::to_string(static_cast<const object*>(p.get())->id) : "null"; }
template <class T> std::string fmt(const std::vector<T>& v)
{
    const uint64_t h = hash_bytes(v.data(), v.size() * sizeof(T));
    auto it = st().designs.find(h);
    if (it != st().designs.end()) return it->second;
    std::string s = "[";
    for (size_t i = 0; i < v.size() && i < 64; ++i) s += (i ? "," : "") + fmt(v[i]);
    if (v.size() > 64) s += ",...(" + std::to_string(v.size()) + ")";
    return s + "]";
}
inline std::string join() { return ""; }
template <class A, class... R> std::string join(const A& a, const R&... r) { return fmt(a) + (sizeof...(R) ? "," + join(r...) : ""); }

template <class... A> int add(const std::string& type, const A&... a)
{
    const int id = st().next++;
    st().lines.push_back("#" + std::to_string(id) + " " + type + "(" + join(a...) + ")");
    return id;
}
template <class... A> void call(int id, const std::string& method, const A&... a)
{
    st().lines.push_back("#" + std::to_string(id) + "." + method + "(" + join(a...) + ")");
}

}  // namespace rec

class io_signature {
public:
    typedef std::shared_ptr<io_signature> sptr;
    static sptr make(int, int, int) { return sptr(new io_signature); }
    static sptr make2(int, int, int, int) { return sptr(new io_signature); }
    static sptr make3(int, int, int, int, int) { return sptr(new io_signature); }
    static sptr makev(int, int, const std::vector<int>&) { return sptr(new io_signature); }
};

class basic_block : public rec::object {
public:
    // the setters the hier blocks call on their members
    template <class T> void set_taps(const std::vector<T>& t) { rec::call(id, "set_taps", t); }
    void set_k(double k) { rec::call(id, "set_k", k); }
    void set_k(gr_complex k) { rec::call(id, "set_k", k); }
    void set_gain(double v) { rec::call(id, "set_gain", v); }
    void set_threshold(double v) { rec::call(id, "set_threshold", v); }
    void set_max_gain(double v) { rec::call(id, "set_max_gain", v); }
    void set_frequency(double v) { rec::call(id, "set_frequency", v); }
    void set_attack_rate(double v) { rec::call(id, "set_attack_rate", v); }
    void set_decay_rate(double v) { rec::call(id, "set_decay_rate", v); }
    void set_sensitivity(double v) { rec::call(id, "set_sensitivity", v); }
    void declare_sample_delay(int, int d) { rec::call(id, "declare_sample_delay", d); }
    void declare_sample_delay(unsigned d) { rec::call(id, "declare_sample_delay", d); }
    void set_thread_priority(int) {}
    void set_tag_propagation_policy(int) {}
    void set_min_output_buffer(long) {}
    void set_max_output_buffer(long) {}
};
typedef std::shared_ptr<basic_block> basic_block_sptr;
enum tag_propagation_policy_t { TPP_DONT = 0, TPP_ALL_TO_ALL = 1, TPP_ONE_TO_ONE = 2, TPP_CUSTOM = 3 };
// gr::block and its sync_* forms also carry the work()-side interface of oracle/gr_stub/gnuradio/block.h, so that the reference's OWN
// custom blocks (src/gr/*.cpp) compile against this tree as well; constructed by name they record themselves as "custom::<name>"
struct tag_t {
    uint64_t offset; pmt::pmt_t key, value; unsigned port = 0;
    static bool offset_compare(const tag_t& a, const tag_t& b) { return a.offset < b.offset; }
};
class block : public basic_block {
public:
    enum { WORK_CALLED_PRODUCE = -2, WORK_DONE = -1 };
    enum { TPP_DONT = 0, TPP_ALL_TO_ALL = 1, TPP_ONE_TO_ONE = 2 };
    block() {}
    block(const std::string& name, io_signature::sptr, io_signature::sptr) { id = rec::add("custom::" + name); }
    virtual ~block() {}
    virtual void forecast(int, gr_vector_int&) {}
    virtual int general_work(int, gr_vector_int&, gr_vector_const_void_star&, gr_vector_void_star&) { return 0; }
    void set_relative_rate(double) {}
    void set_alignment(int) {}
    void consume_each(int) {}
    void consume(int, int) {}
    void produce(int, int) {}
    uint64_t nitems_written(unsigned) const { return 0; }
    uint64_t nitems_read(unsigned) const { return 0; }
    void add_item_tag(unsigned, uint64_t, const pmt::pmt_t&, const pmt::pmt_t&) {}
    void get_tags_in_window(std::vector<tag_t>& v, unsigned, uint64_t, uint64_t, const pmt::pmt_t&) { v.clear(); }
    void get_tags_in_window(std::vector<tag_t>& v, unsigned, uint64_t, uint64_t) { v.clear(); }
    void get_tags_in_range(std::vector<tag_t>& v, unsigned, uint64_t, uint64_t, const pmt::pmt_t&) { v.clear(); }
    void get_tags_in_range(std::vector<tag_t>& v, unsigned, uint64_t, uint64_t) { v.clear(); }
    void set_min_noutput_items(int) {}
    void set_max_noutput_items(int) {}
    void set_history(unsigned) {}
    unsigned history() const { return 1; }
    void set_output_multiple(int) {}
    std::vector<tag_t> stub_in_tags, stub_tags;
    long stub_consumed = 0;
    uint64_t stub_written = 0, stub_read = 0;
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
class sync_decimator : public sync_block {
public:
    sync_decimator(const std::string& name, io_signature::sptr a, io_signature::sptr b, unsigned) : sync_block(name, a, b) {}
};

class hier_block2 : public basic_block {
public:
    hier_block2(const std::string& name, io_signature::sptr, io_signature::sptr) { id = 0; rec::st().lines.push_back("hier " + name); d_self.reset(new basic_block); d_self->id = 0; }
    basic_block_sptr self() { return d_self; }
    void connect(basic_block_sptr a, int pa, basic_block_sptr b, int pb)
    { rec::st().lines.push_back("#" + std::to_string(a->id) + ":" + std::to_string(pa) + " -> #" + std::to_string(b->id) + ":" + std::to_string(pb)); }
    void disconnect(basic_block_sptr, int, basic_block_sptr, int) {}
    void lock() {} void unlock() {}
private:
    basic_block_sptr d_self;
};

enum endianness_t { GR_MSB_FIRST = 0, GR_LSB_FIRST = 1 };

}  // namespace gr

namespace gnuradio { template <class T> std::shared_ptr<T> get_initial_sptr(T* p) { return std::shared_ptr<T>(p); } }

// a block class: ns::cls with sptr and a variadic recording make()
#define GR_REC_BLOCK(NS, CLS)                                                                                          \
    namespace gr { namespace NS { class CLS : public gr::block {                                                        \
    public:                                                                                                            \
        typedef std::shared_ptr<CLS> sptr;                                                                             \
        template <class... A> static sptr make(const A&... a) { sptr p(new CLS); p->id = gr::rec::add(#NS "::" #CLS, a...); return p; } \
    }; } }
