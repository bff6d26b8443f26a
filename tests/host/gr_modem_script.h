// gr_modem_script.h -- TEST INFRASTRUCTURE.  ONE driver source for TWO classes named gr_modem: the reference's own (src/gr_modem.cpp compiled where it
// lies, oracle/ref_driver_modem.cpp -> oracle/_ref/gr_modem_script_ref) and the HIP path's (qradiolink_amd/host/qt/gr_modem.*,
// tests/host/test_gr_modem_literal.cpp).  Include it AFTER the gr_modem.h under test.  It uses nothing but the reference's public interface
// (src/gr_modem.h:55-139): the slots radiocontroller.cpp calls and the signals it connects (src/radiocontroller.cpp:121-152, 1298, 1969-2078).
//   * the signals -- moc's job in a real build -- are defined HERE as recorders, identically for both classes: "S <signal> <arguments>";
//   * script_setup / script_transmit call the TX slots in a fixed order;  script_poll calls demodulate() once and logs "R <return value>".
// The two logs must be equal line by line (tests/test_gpu_modem_literal.py).
#pragma once
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

static FILE* g_script_log = nullptr;
static std::string script_hex(const unsigned char* p, int n)
{
    static const char* d = "0123456789abcdef";
    std::string s;
    for (int i = 0; i < n; ++i) { s.push_back(d[p[i] >> 4]); s.push_back(d[p[i] & 15]); }
    return s;
}
static void script_log(const std::string& s) { if (g_script_log) { std::fputs(s.c_str(), g_script_log); std::fputc('\n', g_script_log); } }

// ---- the signals (what moc generates; a slot connected to one of the buffer-carrying signals owns the buffer, as in radiocontroller.cpp)
void gr_modem::pcmAudio(std::vector<float>* pcm) { script_log("S pcmAudio " + std::to_string(pcm->size())); delete pcm; }
void gr_modem::digitalAudio(unsigned char* c2data, int size) { script_log("S digitalAudio " + script_hex(c2data, size)); delete[] c2data; }
void gr_modem::videoData(unsigned char* d, int size) { script_log("S videoData " + script_hex(d, size)); delete[] d; }
void gr_modem::netData(unsigned char* d, int size) { script_log("S netData " + script_hex(d, size)); delete[] d; }
void gr_modem::demodulated_audio(short*, short) { script_log("S demodulated_audio"); }
void gr_modem::textReceived(QString text, bool html)
{
    const std::string t = text.toStdString();
    script_log(std::string("S textReceived ") + (html ? "html " : "plain ") + script_hex(reinterpret_cast<const unsigned char*>(t.data()), (int)t.size()));
}
void gr_modem::protoReceived(QByteArray data) { script_log("S protoReceived " + script_hex(reinterpret_cast<const unsigned char*>(data.constData()), data.size())); }
void gr_modem::callsignReceived(QString text) { script_log("S callsignReceived " + text.toStdString()); }
void gr_modem::m17FrameInfoReceived(QString src, QString dest, uint16_t CAN) { script_log("S m17FrameInfoReceived " + src.toStdString() + " " + dest.toStdString() + " " + std::to_string(CAN)); }
void gr_modem::audioFrameReceived() { script_log("S audioFrameReceived"); }
void gr_modem::dataFrameReceived() { script_log("S dataFrameReceived"); }
void gr_modem::syncIssues() { script_log("S syncIssues"); }
void gr_modem::receiveEnd() { script_log("S receiveEnd"); }
void gr_modem::endAudioTransmission() { script_log("S endAudioTransmission"); }
void gr_modem::endBeep() { script_log("S endBeep"); }

// ---- the script
static void script_setup(gr_modem& m, int mode)
{
    // the order radiocontroller.cpp uses: initTX / initRX (:1969-2020), the GUI taps off, the transmitter's gain
    m.initTX(mode, 433500000, "", "", 0);
    m.initRX(mode, "", "", 0);
    m.setCarrierOffset(0);
    m.setTxCarrierOffset(0);
    m.enableGUIConst(false);
    m.enableGUIFFT(false);
    m.enableRSSI(false);
    m.setBbGain(5);          // 5 / 5.0 = 1.0
    m.startRX();
    m.startTX();
}
static int script_frame_length(int mode)   // the payload bytes a voice frame of the mode carries (what the codec hands to transmitDigitalAudio)
{
    switch (mode) {
    case 1: case 4: case 15: case 22: return 47;       // QPSK20K, 4FSK10KFM, 2FSK10KFM, GMSK10K
    case 6: case 16: case 18: case 21: case 24: return 4;   // the 1k modes
    case 26: return 1516;                                // QPSK250K
    case 27: return 622;                                 // 4FSK100K
    default: return 7;
    }
}
static void script_transmit(gr_modem& m, int mode, int nframes, const char* callsign)
{
    const int L = script_frame_length(mode);
    m.startTransmission(QString(callsign));
    const bool fast = mode == 26 || mode == 27;   // QPSK250K / 4FSK100K: the IP / video modes -- their receivers know the IP, video and end sync words only (src/gr_modem.cpp:1211-1237)
    for (int f = 0; f < nframes; ++f) {
        unsigned char* d = new unsigned char[L];
        for (int i = 0; i < L; ++i) d[i] = (unsigned char)(31 * f + 7 * i + 1);
        if (!fast) m.transmitDigitalAudio(d, L);          // takes ownership
        else if (f & 1) m.transmitNetData(d, L);
        else m.transmitVideoData(d, L);
    }
    if (L >= 7 && !fast) {
        m.transmitTextData(QString("literal boundary: the quick brown fox"));
        std::vector<char> bin((size_t)L + 3);
        for (size_t i = 0; i < bin.size(); ++i) bin[i] = (char)(200 - i);
        m.transmitBinData(QByteArray(bin.data(), (int)bin.size()));
    }
    m.endTransmission(QString(callsign));
}
static bool script_poll(gr_modem& m)
{
    const bool r = m.demodulate();
    script_log(std::string("R ") + (r ? "1" : "0"));
    return r;
}
