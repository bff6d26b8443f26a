// qt_stub shadow of src/gr/gr_mod_base.h: records the bytes gr_modem hands to the byte source
#pragma once
#include <string>
#include <vector>
#include <QMap>
#include <QVector>
#include "src/bursttimer.h"
#include "src/DMR/dmrtiming.h"
#include "src/DMR/dmrframe.h"
class gr_mod_base {
public:
    gr_mod_base(BurstTimer*, DMRTiming*, void* = nullptr, int64_t = 0, float = 0, std::string = "", std::string = "", int = 0, int = 3, int = 25000) {}
    int set_data(std::vector<unsigned char>* d) { sent.insert(sent.end(), d->begin(), d->end()); calls++; delete d; return 0; }
    int set_audio(std::vector<float>* a) { delete a; return 0; }
    int setDMRData(std::vector<DMRFrame>&) { return 0; }
    void set_mode(int m) { mode = m; }
    void start(int = 0) {} void stop() {} void tune(int64_t) {} void set_samp_rate(int) {} void set_power(float, std::string = "") {}
    void set_filter_width(int, int) {} void set_cw_k(bool) {} void set_ctcss(float) {} void set_carrier_offset(int64_t) {} void set_bb_gain(float) {}
    int64_t reset_carrier_offset() { return 0; } void flush_sources() {}
    const QMap<std::string, QVector<int>> get_gain_names() const { return QMap<std::string, QVector<int>>(); }
    std::vector<unsigned char> sent; int calls = 0; int mode = -1;
};
