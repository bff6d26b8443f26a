// gr_stub — TEST INFRASTRUCTURE: generic (scalar, in-order) forms of the VOLK kernels the reference's cessb blocks call, so that
// clipper_cc_impl.cc / stretcher_cc_impl.cc compile unmodified.  cos / sin are the oracle's deterministic polynomial (the declared
// substitution of this repository: VOLK's own approximations differ per SIMD path), magnitude is sqrtf(re^2 + im^2).
#pragma once
#include <cmath>
#include <complex>
extern "C" void orc_sincosf(float x, float* s, float* c);
typedef std::complex<float> lv_32fc_t;
inline size_t volk_get_alignment() { return 32; }
inline void volk_32fc_magnitude_32f(float* out, const lv_32fc_t* in, unsigned n) { for (unsigned i = 0; i < n; ++i) { const float a = in[i].real() * in[i].real(), b = in[i].imag() * in[i].imag(); out[i] = sqrtf(a + b); } }
inline void volk_32f_x2_min_32f(float* out, const float* a, const float* b, unsigned n) { for (unsigned i = 0; i < n; ++i) out[i] = a[i] < b[i] ? a[i] : b[i]; }
inline void volk_32f_x2_max_32f(float* out, const float* a, const float* b, unsigned n) { for (unsigned i = 0; i < n; ++i) out[i] = a[i] > b[i] ? a[i] : b[i]; }
inline void volk_32f_cos_32f(float* out, const float* in, unsigned n) { for (unsigned i = 0; i < n; ++i) { float s, c; orc_sincosf(in[i], &s, &c); out[i] = c; } }
inline void volk_32f_sin_32f(float* out, const float* in, unsigned n) { for (unsigned i = 0; i < n; ++i) { float s, c; orc_sincosf(in[i], &s, &c); out[i] = s; } }
inline void volk_32f_x2_multiply_32f(float* out, const float* a, const float* b, unsigned n) { for (unsigned i = 0; i < n; ++i) out[i] = a[i] * b[i]; }
inline void volk_32f_x2_add_32f(float* out, const float* a, const float* b, unsigned n) { for (unsigned i = 0; i < n; ++i) out[i] = a[i] + b[i]; }
inline void volk_32f_x2_subtract_32f(float* out, const float* a, const float* b, unsigned n) { for (unsigned i = 0; i < n; ++i) out[i] = a[i] - b[i]; }
inline void volk_32f_x2_divide_32f(float* out, const float* a, const float* b, unsigned n) { for (unsigned i = 0; i < n; ++i) out[i] = a[i] / b[i]; }
inline void volk_32f_s32f_multiply_32f(float* out, const float* a, float k, unsigned n) { for (unsigned i = 0; i < n; ++i) out[i] = a[i] * k; }
inline void volk_32f_x2_interleave_32fc(lv_32fc_t* out, const float* re, const float* im, unsigned n) { for (unsigned i = 0; i < n; ++i) out[i] = lv_32fc_t(re[i], im[i]); }
inline void volk_32fc_deinterleave_real_32f(float* out, const lv_32fc_t* in, unsigned n) { for (unsigned i = 0; i < n; ++i) out[i] = in[i].real(); }
inline void volk_32fc_deinterleave_imag_32f(float* out, const lv_32fc_t* in, unsigned n) { for (unsigned i = 0; i < n; ++i) out[i] = in[i].imag(); }
