// firdes.hpp — host-side filter design, loop gains and lookup tables the HIP kernels are loaded with.
// Mirrors what the reference obtains from gr::filter::firdes / gr::fft::window / control loops of
// GNU Radio 3.10 (call sites: gr_demod_2fsk.cpp:82-110, gr_demod_gmsk.cpp:80-98,
// gr_demod_qpsk.cpp:92-112, gr_demod_base.cpp:1333-1336).  Design math runs in double and is
// rounded to float where upstream stores float.
#pragma once
#include <complex>
#include <cstdint>
#include <vector>

namespace qrl {

enum Window { WIN_HAMMING = 0, WIN_HANN = 1, WIN_BLACKMAN = 2, WIN_RECTANGULAR = 3, WIN_BLACKMAN_HARRIS = 5 };

std::vector<float> window(Window type, int ntaps);
int compute_ntaps(double fs, double tw, Window w);
int compute_ntaps_windes(double fs, double tw, double atten_db);
std::vector<float> low_pass(double gain, double fs, double fc, double tw, Window w = WIN_HAMMING);
// matched filter of dsss_decoder_cc (reference src/gr/dsss_decoder_cc_impl.cc:41-107): Barker-13, reversed, sps samples per chip, through
// RRC(1, sps, 1, 0.35, 11 sps); 13 sps + 11 sps real taps (== oracle orc_dsss_taps, bit for bit)
std::vector<float> dsss_matched_filter(int sps);
std::vector<float> low_pass_2(double gain, double fs, double fc, double tw, double atten_db, Window w = WIN_HAMMING);
std::vector<std::complex<float>> complex_band_pass(double gain, double fs, double lo, double hi, double tw, Window w = WIN_HAMMING);
std::vector<float> band_pass_2(double gain, double fs, double lo, double hi, double tw, double atten_db, Window w = WIN_HAMMING);
std::vector<std::complex<float>> complex_band_pass_2(double gain, double fs, double lo, double hi, double tw, double atten_db, Window w = WIN_HAMMING);
// reference src/gr/emphasis.cpp:16-43 (gr-analog fm_emph.py): b = {b0, b0}, a = {1, -p1}
void deemph_taps(int sample_rate, double tau, double a[2], double b[2]);
// reference src/gr/emphasis.cpp:44-89 with its default upper corner fh = 0.925 fs / 2: b = {g b0, -g b0 z1}, a = {1, -p1}
void preemph_taps(int sample_rate, double tau, double a[2], double b[2]);
// squelch_base_cc ramp envelope 0.5 - cos(pi k / ramp) / 2, k = 0 .. ramp
std::vector<float> squelch_envelope(int ramp);
std::vector<float> root_raised_cosine(double gain, double fs, double symrate, double alpha, int ntaps);
std::vector<float> gaussian(double gain, double spb, double bt, int ntaps);

// fll_band_edge_cc design_filter: returns taps in the order the filter applies them,
// T[j] multiplies y[n-j] (upstream stores them reversed and reverses again in the FIR)
void fll_band_edge_taps(float sps, float rolloff, int n, std::vector<std::complex<float>>& lower,
                        std::vector<std::complex<float>>& upper);
void control_loop_gains(float bw, float& alpha, float& beta);
void clock_loop_gains(float loop_bw, float zeta, float ted_gain, float& alpha, float& beta);

std::vector<float> mmse_table();  // 129 x 8
std::vector<float> atan_table();  // 257
std::vector<float> tanh_table();  // 256

uint64_t phase_inc_to_turn(double radians_per_sample);
// host evaluation of the NCO polynomial (same arithmetic as the device function)
void sincos_turn_host(uint64_t angle, float& s, float& c);

}  // namespace qrl
