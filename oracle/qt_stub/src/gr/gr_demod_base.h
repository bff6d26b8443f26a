// qt_stub shadow of src/gr/gr_demod_base.h: the mailboxes gr_modem polls, fed by the test; everything else is a no-op
#pragma once
#include <complex>
#include <deque>
#include <string>
#include <vector>
#include <QMap>
#include <QVector>
#include "src/bursttimer.h"
#include "src/DMR/dmrtiming.h"
#include "src/DMR/dmrframe.h"
typedef std::complex<float> gr_complex;
class gr_demod_base {
public:
    gr_demod_base(BurstTimer*, DMRTiming*, void* = nullptr, double = 0, float = 0, std::string = "", std::string = "", int = 0, int = 3, int = 25000) {}
    std::vector<unsigned char>* getData() { return take(0); }
    std::vector<unsigned char>* getData(int nr) { return take(nr); }
    std::vector<float>* getAudio() { return nullptr; }
    std::vector<DMRFrame> getDMRData(bool = false) { return std::vector<DMRFrame>(); }
    std::vector<gr_complex>* get_constellation_data() { return nullptr; }
    void get_FFT_data(float*, unsigned int& n) { n = 0; }
    void get_sample_data(float*, unsigned int& n) { n = 0; }
    float get_rssi() { return 0.0f; }
    double get_freq() { return 0.0; }
    const QMap<std::string, QVector<int>> get_gain_names() const { return QMap<std::string, QVector<int>>(); }
    void set_mode(int m) { mode = m; }
    void start(int = 0) {} void stop() {} void tune(int64_t) {}
    void set_time_sink_samp_rate(int) {} void set_time_domain_filter_width(double) {} void set_squelch(int) {} void set_sample_window(unsigned) {}
    void set_samp_rate(int) {} void set_rx_sensitivity(double, std::string = "") {} void set_gain(float) {} void set_filter_width(int, int) {}
    void set_fft_size(int) {} void set_ctcss(float) {} void set_carrier_offset(int64_t) {} void set_agc_decay(int) {} void set_agc_attack(int) {}
    void enable_time_domain(bool) {} void enable_rssi(bool) {} void enable_gui_fft(bool) {} void enable_gui_const(bool) {} void enable_demodulator(bool) {}
    void calibrate_rssi(float) {}
    // test side
    std::deque<std::vector<unsigned char>*> q[3];
    int mode = -1;
private:
    std::vector<unsigned char>* take(int nr) { if (q[nr].empty()) return nullptr; auto* v = q[nr].front(); q[nr].pop_front(); return v; }
};
