// gr_compat.h — the handful of GNU Radio 3.10 runtime types the QRadioLink blocks are written against
// (reference src/gr/gr_4fsk_discriminator.h:19-21, src/gr/gr_bit_sink.h), so that the HIP adaptor blocks
// compile in a tree without GNU Radio.  With GNU Radio installed the real headers are used instead.
#pragma once
#if __has_include(<gnuradio/sync_block.h>)
#include <gnuradio/io_signature.h>
#include <gnuradio/sync_block.h>
#include <gnuradio/sync_interpolator.h>
#include <gnuradio/thread/thread.h>
#else
#include <complex>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
typedef std::complex<float> gr_complex;
typedef std::vector<const void*> gr_vector_const_void_star;
typedef std::vector<void*> gr_vector_void_star;
typedef std::vector<int> gr_vector_int;
namespace gr {
class io_signature {
public:
    typedef std::shared_ptr<io_signature> sptr;
    static sptr make(int min_streams, int max_streams, int sizeof_stream_item)
    { return sptr(new io_signature(min_streams, max_streams, sizeof_stream_item)); }
    int min_streams() const { return d_min; }
    int max_streams() const { return d_max; }
    int sizeof_stream_item(int) const { return d_size; }
private:
    io_signature(int a, int b, int c) : d_min(a), d_max(b), d_size(c) {}
    int d_min, d_max, d_size;
};
class sync_block {
public:
    sync_block(const std::string& name, io_signature::sptr in, io_signature::sptr out) : d_name(name), d_in(in), d_out(out) {}
    virtual ~sync_block() {}
    // same contract as gr::sync_block::work: returns the number of output items produced (== input items consumed)
    virtual int work(int noutput_items, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items) = 0;
    const std::string& name() const { return d_name; }
    io_signature::sptr input_signature() const { return d_in; }
    io_signature::sptr output_signature() const { return d_out; }
    void set_output_multiple(int m) { d_multiple = m; }
    int output_multiple() const { return d_multiple; }
private:
    std::string d_name; io_signature::sptr d_in, d_out; int d_multiple = 1;
};
class sync_interpolator : public sync_block {
public:
    sync_interpolator(const std::string& name, io_signature::sptr in, io_signature::sptr out, unsigned interpolation)
        : sync_block(name, in, out), d_interp(interpolation) {}
    unsigned interpolation() const { return d_interp; }
private:
    unsigned d_interp;
};
namespace thread {
typedef std::mutex mutex;
typedef std::lock_guard<std::mutex> scoped_lock;
}  // namespace thread
}  // namespace gr
#endif
