// gr_stub — TEST INFRASTRUCTURE: gr::filter::kernel::fir_filter_{ccf,ccc} for the reference's dsss_decoder_cc_impl.cc.
// filter(in) = sum_k taps[k] in[ntaps - 1 - k] (GNU Radio stores the taps reversed and takes a dot product with in[0 .. ntaps));
// the sum is the oracle's fmaf chain, k ascending (VOLK leaves the order to the machine), so values can be compared exactly.
#pragma once
#include <cmath>
#include <complex>
#include <vector>

#include <gnuradio/block.h>

namespace gr {
namespace filter {
namespace kernel {

class fir_filter_ccf {
public:
    explicit fir_filter_ccf(const std::vector<float>& taps) : d_taps(taps) {}
    gr_complex filter(const gr_complex* in) const
    {
        float re = 0.0f, im = 0.0f;
        const int nt = (int)d_taps.size();
        for (int k = 0; k < nt; ++k) { re = fmaf(d_taps[k], in[nt - 1 - k].real(), re); im = fmaf(d_taps[k], in[nt - 1 - k].imag(), im); }
        return gr_complex(re, im);
    }
private:
    std::vector<float> d_taps;
};
class fir_filter_ccc {
public:
    explicit fir_filter_ccc(const std::vector<gr_complex>& taps) : d_taps(taps) {}
    // (the only instance here has real-valued taps stored as complex: the imaginary products are exact zeros)
    gr_complex filter(const gr_complex* in) const
    {
        float re = 0.0f, im = 0.0f;
        const int nt = (int)d_taps.size();
        for (int k = 0; k < nt; ++k) { re = fmaf(d_taps[k].real(), in[nt - 1 - k].real(), re); im = fmaf(d_taps[k].real(), in[nt - 1 - k].imag(), im); }
        return gr_complex(re, im);
    }
private:
    std::vector<gr_complex> d_taps;
};

}  // namespace kernel
}  // namespace filter
}  // namespace gr
