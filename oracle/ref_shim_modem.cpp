// ref_shim_modem.cpp -- TEST INFRASTRUCTURE.  The reference's OWN gr_modem class (/root/reference/src/gr_modem.cpp: toggleRxMode /
// toggleTxMode mode tables, demodulate, synchronize, findSync, processReceivedData, frame, transmit, sendCallsign, start /
// endTransmission, transmit*Data) compiled unmodified against oracle/qt_stub (a sliver of Qt; shadows of the project classes gr_modem
// only forwards to).  The Qt signals -- which moc would generate -- are defined here and record their arguments; the stub
// gr_demod_base hands out the bit vectors the test queues, the stub gr_mod_base records the bytes handed to the byte source.
// Pins: oracle orc_modem_sync (the k_framesync contract) and host/gr_modem_hip.cpp (tests/test_ref_modem.py, test_gpu_modem_facade.py).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>
#include <array>
#include <bitset>
#include <complex>
#include <chrono>
#include <mutex>
#include <map>
#include <deque>
#include <algorithm>
#include <memory>
#include <experimental/array>

#define private public
#include "src/gr_modem.h"
#undef private

static std::vector<std::string>* g_events = nullptr;
static std::string hex(const unsigned char* p, int n)
{
    static const char* d = "0123456789abcdef";
    std::string s;
    for (int i = 0; i < n; ++i) { s.push_back(d[p[i] >> 4]); s.push_back(d[p[i] & 15]); }
    return s;
}
static void ev(const std::string& s) { if (g_events) g_events->push_back(s); }

// ---- the signals (moc output in a real build): record, and free what a slot would free
void gr_modem::pcmAudio(std::vector<float>* pcm) { ev("pcm " + std::to_string(pcm->size())); delete pcm; }
void gr_modem::digitalAudio(unsigned char* c2data, int size) { ev("audio " + hex(c2data, size)); delete[] c2data; }
void gr_modem::videoData(unsigned char* d, int size) { ev("video " + hex(d, size)); delete[] d; }
void gr_modem::netData(unsigned char* d, int size) { ev("net " + hex(d, size)); delete[] d; }
void gr_modem::demodulated_audio(short*, short) {}
void gr_modem::textReceived(QString text, bool html) { ev(std::string(html ? "html " : "text ") + hex(reinterpret_cast<const unsigned char*>(text.d.data()), (int)text.d.size())); }
void gr_modem::protoReceived(QByteArray data) { ev("proto " + hex(reinterpret_cast<const unsigned char*>(data.d.data()), (int)data.d.size())); }
void gr_modem::callsignReceived(QString text) { ev("callsign " + text.d); }
void gr_modem::m17FrameInfoReceived(QString src, QString dest, uint16_t CAN) { ev("m17info " + src.d + " " + dest.d + " " + std::to_string(CAN)); }
void gr_modem::audioFrameReceived() { ev("audioframe"); }
void gr_modem::dataFrameReceived() { ev("dataframe"); }
void gr_modem::syncIssues() { ev("syncissues"); }
void gr_modem::receiveEnd() { ev("receiveend"); }
void gr_modem::endAudioTransmission() { ev("endaudio"); }
void gr_modem::endBeep() { ev("endbeep"); }

struct RefModem {
    Settings settings; Logger logger; DMRControl dmr;
    gr_modem* m = nullptr;
    std::vector<std::string> events;
};

extern "C" {

void* ref_modem_new(void)
{
    RefModem* r = new RefModem;
    r->m = new gr_modem(&r->settings, &r->logger, &r->dmr);
    return r;
}
void ref_modem_free(void* p) { RefModem* r = static_cast<RefModem*>(p); g_events = nullptr; delete r->m; delete r; }
void ref_modem_init_rx(void* p, int mode) { RefModem* r = static_cast<RefModem*>(p); g_events = &r->events; r->m->initRX(mode, "", "", 0); }
void ref_modem_init_tx(void* p, int mode) { RefModem* r = static_cast<RefModem*>(p); g_events = &r->events; r->m->initTX(mode, 433500000, "", "", 0); }
void ref_modem_toggle_rx(void* p, int mode) { RefModem* r = static_cast<RefModem*>(p); g_events = &r->events; r->m->toggleRxMode(mode); }
void ref_modem_toggle_tx(void* p, int mode) { RefModem* r = static_cast<RefModem*>(p); g_events = &r->events; r->m->toggleTxMode(mode); }
int ref_modem_rx_frame_length(void* p) { return static_cast<RefModem*>(p)->m->_rx_frame_length; }
int ref_modem_tx_frame_length(void* p) { return static_cast<RefModem*>(p)->m->_tx_frame_length; }
int ref_modem_bit_buf_len(void* p) { return static_cast<RefModem*>(p)->m->_bit_buf_len; }
// one vector for gr_demod_base::getData() (nr 0) / getData(1) / getData(2)
void ref_modem_push(void* p, int nr, const uint8_t* bits, size_t n)
{
    RefModem* r = static_cast<RefModem*>(p);
    r->m->_gr_demod_base->q[nr].push_back(new std::vector<unsigned char>(bits, bits + n));
}
int ref_modem_demodulate(void* p) { RefModem* r = static_cast<RefModem*>(p); g_events = &r->events; return r->m->demodulate() ? 1 : 0; }
// newline separated events since the last call
size_t ref_modem_events(void* p, char* buf, size_t cap)
{
    RefModem* r = static_cast<RefModem*>(p);
    std::string s;
    for (const std::string& e : r->events) { s += e; s.push_back('\n'); }
    r->events.clear();
    const size_t n = s.size() < cap ? s.size() : cap;
    std::memcpy(buf, s.data(), n);
    return n;
}
// TX API
void ref_modem_start_tx(void* p, const char* callsign) { RefModem* r = static_cast<RefModem*>(p); g_events = &r->events; r->m->startTransmission(QString(callsign)); }
void ref_modem_end_tx(void* p, const char* callsign) { RefModem* r = static_cast<RefModem*>(p); g_events = &r->events; r->m->endTransmission(QString(callsign)); }
void ref_modem_send_callsign(void* p, const char* callsign) { static_cast<RefModem*>(p)->m->sendCallsign(QString(callsign)); }
void ref_modem_tx_audio(void* p, const uint8_t* d, int n) { unsigned char* c = new unsigned char[n]; std::memcpy(c, d, (size_t)n); static_cast<RefModem*>(p)->m->transmitDigitalAudio(c, n); }
void ref_modem_tx_video(void* p, const uint8_t* d, int n) { unsigned char* c = new unsigned char[n]; std::memcpy(c, d, (size_t)n); static_cast<RefModem*>(p)->m->transmitVideoData(c, n); }
void ref_modem_tx_net(void* p, const uint8_t* d, int n) { unsigned char* c = new unsigned char[n]; std::memcpy(c, d, (size_t)n); static_cast<RefModem*>(p)->m->transmitNetData(c, n); }
void ref_modem_tx_text(void* p, const uint8_t* d, int n, int frame_type) { static_cast<RefModem*>(p)->m->transmitTextData(QString(std::string(reinterpret_cast<const char*>(d), (size_t)n)), frame_type); }
void ref_modem_tx_bin(void* p, const uint8_t* d, int n, int frame_type) { static_cast<RefModem*>(p)->m->transmitBinData(QByteArray(reinterpret_cast<const char*>(d), n), frame_type); }
// the bytes gr_mod_base::set_data received since the last call
size_t ref_modem_tx_take(void* p, uint8_t* buf, size_t cap)
{
    RefModem* r = static_cast<RefModem*>(p);
    std::vector<unsigned char>& s = r->m->_gr_mod_base->sent;
    const size_t n = s.size() < cap ? s.size() : cap;
    std::memcpy(buf, s.data(), n);
    s.clear();
    return n;
}

}
