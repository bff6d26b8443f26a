// gr_stub — TEST INFRASTRUCTURE: gr::filter::firdes::root_raised_cosine through the oracle's restatement (liborc.so)
#pragma once
#include <vector>
extern "C" int orc_root_raised_cosine(double gain, double fs, double symrate, double alpha, int ntaps, float* taps);
namespace gr {
namespace filter {
struct firdes {
    static std::vector<float> root_raised_cosine(double gain, double fs, double symrate, double alpha, int ntaps)
    {
        std::vector<float> t((size_t)orc_root_raised_cosine(gain, fs, symrate, alpha, ntaps, nullptr));
        orc_root_raised_cosine(gain, fs, symrate, alpha, ntaps, t.data());
        return t;
    }
};
}  // namespace filter
}  // namespace gr
