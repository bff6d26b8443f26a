// rec_stub — TEST INFRASTRUCTURE: gr::filter::firdes as a recorder: every design call returns a 4-tap token vector whose hash is registered
// with the text of the call, so that a later ::make(..., taps) prints "low_pass(1,1000000,10000,10000,enum:5)" instead of numbers
#pragma once
#include <gnuradio/recording.h>
#include <gnuradio/rec_enums.h>
namespace gr {
namespace filter {
struct firdes {
    template <class T, class... A> static std::vector<T> design(const char* name, const A&... a)
    {
        const std::string text = std::string(name) + "(" + rec::join(a...) + ")";
        const uint64_t h = rec::hash_bytes(text.data(), text.size());
        std::vector<T> v(4);
        for (int i = 0; i < 4; ++i) v[i] = T((float)((h >> (16 * i)) & 0xFFFF) + 0.5f);
        rec::st().designs[rec::hash_bytes(v.data(), v.size() * sizeof(T))] = text;
        return v;
    }
    static std::vector<float> low_pass(double g, double fs, double fc, double tw, fft::window::win_type w = fft::window::WIN_HAMMING, double beta = 6.76)
    { (void)beta; return design<float>("low_pass", g, fs, fc, tw, w); }
    static std::vector<float> low_pass_2(double g, double fs, double fc, double tw, double att, fft::window::win_type w = fft::window::WIN_HAMMING, double beta = 6.76)
    { (void)beta; return design<float>("low_pass_2", g, fs, fc, tw, att, w); }
    static std::vector<float> band_pass(double g, double fs, double lo, double hi, double tw, fft::window::win_type w = fft::window::WIN_HAMMING, double beta = 6.76)
    { (void)beta; return design<float>("band_pass", g, fs, lo, hi, tw, w); }
    static std::vector<float> band_pass_2(double g, double fs, double lo, double hi, double tw, double att, fft::window::win_type w = fft::window::WIN_HAMMING, double beta = 6.76)
    { (void)beta; return design<float>("band_pass_2", g, fs, lo, hi, tw, att, w); }
    static std::vector<gr_complex> complex_band_pass(double g, double fs, double lo, double hi, double tw, fft::window::win_type w = fft::window::WIN_HAMMING, double beta = 6.76)
    { (void)beta; return design<gr_complex>("complex_band_pass", g, fs, lo, hi, tw, w); }
    static std::vector<gr_complex> complex_band_pass_2(double g, double fs, double lo, double hi, double tw, double att, fft::window::win_type w = fft::window::WIN_HAMMING, double beta = 6.76)
    { (void)beta; return design<gr_complex>("complex_band_pass_2", g, fs, lo, hi, tw, att, w); }
    static std::vector<float> root_raised_cosine(double g, double fs, double sr, double alpha, int ntaps)
    { return design<float>("root_raised_cosine", g, fs, sr, alpha, ntaps); }
    static std::vector<float> gaussian(double g, double spb, double bt, int ntaps) { return design<float>("gaussian", g, spb, bt, ntaps); }
};
}  // namespace filter
}  // namespace gr
