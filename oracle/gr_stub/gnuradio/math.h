// gr_stub — TEST INFRASTRUCTURE: gr::fast_atan2f through the oracle's restatement of GNU Radio's table version
#pragma once
#include <complex>
extern "C" float orc_fast_atan2f(float y, float x);
namespace gr {
inline float fast_atan2f(float y, float x) { return orc_fast_atan2f(y, x); }
inline float fast_atan2f(std::complex<float> z) { return orc_fast_atan2f(z.imag(), z.real()); }
}
