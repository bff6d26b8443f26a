// gr_modem.cpp -- the reference's `class gr_modem` interface over the HIP path (see gr_modem.h in this directory).
// Every method cites the reference method it stands for (/root/reference/src/gr_modem.cpp); the protocol logic itself (framing, frame
// synchroniser, frame dispatch) lives in qrl_host::gr_modem_hip, the three-class facade above the C ABI -- this file only gives it the
// reference's names, Qt types and signals, for ONE radio (stream 0 of a one-stream handle).
#include "gr_modem.h"

#include <cstring>
#include <stdexcept>

#include "../gr_modem_hip.h"

using namespace qrl_host;

gr_modem::gr_modem(const Settings *settings, Logger *logger, DMRControl *dmrcontrol, QObject *parent) :   // src/gr_modem.cpp:21-47
    QObject(parent), _settings(settings), _logger(logger), _dmr_control(dmrcontrol), _gr_demod_base(nullptr), _gr_mod_base(nullptr),
    _modem(nullptr), _modem_type_rx(0 /* ModemTypeBPSK2K */), _modem_type_tx(0), _device(0), _rx_max(65536), _tx_max(4096),
    _samp_rate(1000000), _rx_freq(433500000), _tx_freq(433500000), _rx_offset(0), _tx_offset(0), _rx_running(true), _tx_running(true),
    _warned_protocol(false)
{
}

gr_modem::~gr_modem()   // :49-59 (deinitRX / deinitTX without rebuilding the facade object in between)
{
    delete _modem;
    _modem = nullptr;
    if (_gr_demod_base) { _gr_demod_base->stop(); delete _gr_demod_base; _gr_demod_base = nullptr; }
    delete _gr_mod_base;
    _gr_mod_base = nullptr;
}

void gr_modem::setDevice(int device, size_t rx_max_samples, size_t tx_max_bytes)
{
    if (_gr_demod_base || _gr_mod_base) throw std::logic_error("gr_modem::setDevice after initRX / initTX");
    _device = device; _rx_max = rx_max_samples & ~(size_t)1; _tx_max = tx_max_bytes;
}

gr_demod_base_hip *gr_modem::createDemodBase(qrl_runtime &rt, int samp_rate, double offset, size_t max_samples)
{
    return new gr_demod_base_hip(rt, 1, samp_rate, offset, max_samples);
}
gr_mod_base_hip *gr_modem::createModBase(qrl_runtime &rt, int samp_rate, double offset, size_t max_bytes)
{
    return new gr_mod_base_hip(rt, 1, samp_rate, offset, max_bytes);
}

// the facade object holds pointers to both base objects: it is rebuilt whenever one of them appears or goes (its RX search state belongs to
// the demodulator that goes with it; the reference keeps that state in gr_modem itself, and initRX / deinitRX reset the graph it refers to)
void gr_modem::rebuildModem()
{
    delete _modem;
    _modem = nullptr;
    if (!_gr_demod_base && !_gr_mod_base) return;
    gr_modem_events ev;
    ev.pcmAudio = [this](int, std::vector<float> *pcm) { emit pcmAudio(pcm); };
    ev.digitalAudio = [this](int, const unsigned char *d, int n) {   // the reference hands heap buffers to its slots (:1376-1400); so does this
        unsigned char *c = new unsigned char[n]; std::memcpy(c, d, (size_t)n); emit digitalAudio(c, n); };
    ev.videoData = [this](int, const unsigned char *d, int n) { unsigned char *c = new unsigned char[n]; std::memcpy(c, d, (size_t)n); emit videoData(c, n); };
    ev.netData = [this](int, const unsigned char *d, int n) { unsigned char *c = new unsigned char[n]; std::memcpy(c, d, (size_t)n); emit netData(c, n); };
    ev.textReceived = [this](int, const std::string &t, bool html) { emit textReceived(QString::fromLocal8Bit(t.data(), (int)t.size()), html); };
    ev.callsignReceived = [this](int, const std::string &c) { emit callsignReceived(QString::fromStdString(c)); };
    ev.protoReceived = [this](int, const std::vector<unsigned char> &d) { emit protoReceived(QByteArray(reinterpret_cast<const char *>(d.data()), (int)d.size())); };
    ev.dataFrameReceived = [this](int) { emit dataFrameReceived(); };
    ev.endAudioTransmission = [this](int) { emit endAudioTransmission(); };
    ev.receiveEnd = [this](int) { emit receiveEnd(); };
    _modem = new gr_modem_hip(_gr_demod_base, _gr_mod_base, ev);
    _modem->set_branch_rule(_reference_branch_rule ? gr_modem_hip::BranchRuleReference : gr_modem_hip::BranchRuleBoth);
    _modem->set_burst_ip_modem(_settings && _settings->burst_ip_modem);
    if (_gr_demod_base) _modem->toggleRxMode(_modem_type_rx);
    if (_gr_mod_base) _modem->toggleTxMode(_modem_type_tx);
}

void gr_modem::initTX(int modem_type, int64_t frequency, std::string, std::string, int, int initial_gain, int, int)   // :61-70
{
    _modem_type_tx = modem_type;
    _tx_freq = frequency;
    if (!_rt) _rt.reset(new qrl_runtime(_device));
    delete _gr_mod_base;
    _gr_mod_base = createModBase(*_rt, _samp_rate, (double)_tx_offset, _tx_max);
    (void)initial_gain;   // the SDR's TX gain (set_power): hardware, behind txSamples()
    rebuildModem();
}

void gr_modem::initRX(int modem_type, std::string, std::string, int, int, int)   // :72-80
{
    _modem_type_rx = modem_type;
    if (!_rt) _rt.reset(new qrl_runtime(_device));
    delete _gr_demod_base;
    _gr_demod_base = createDemodBase(*_rt, _samp_rate, (double)_rx_offset, _rx_max);
    rebuildModem();
}

void gr_modem::deinitTX(int modem_type)   // :82-92
{
    if (!_gr_mod_base) return;
    _modem_type_tx = modem_type;
    delete _gr_mod_base;
    _gr_mod_base = nullptr;
    rebuildModem();
}

void gr_modem::deinitRX(int modem_type)   // :94-103
{
    if (!_gr_demod_base) return;
    _modem_type_rx = modem_type;
    _gr_demod_base->stop();
    delete _gr_demod_base;
    _gr_demod_base = nullptr;
    rebuildModem();
}

void gr_modem::toggleTxMode(int modem_type)   // :105-199 (the frame-length table lives in qrl_host::modem_tx_frame_length)
{
    _modem_type_tx = modem_type;
    if (_modem && _gr_mod_base) _modem->toggleTxMode(modem_type);
}

void gr_modem::toggleRxMode(int modem_type)   // :201-322
{
    _modem_type_rx = modem_type;
    if (_modem && _gr_demod_base) _modem->toggleRxMode(modem_type);
}

// ---- proxy methods (:329-622): forwarded to the two base objects where the path has the function, stored where the function is the SDR's
const QMap<std::string, QVector<int> > gr_modem::getRxGainNames() const { return QMap<std::string, QVector<int> >(); }   // the SDR's gain stages
const QMap<std::string, QVector<int> > gr_modem::getTxGainNames() const { return QMap<std::string, QVector<int> >(); }
void gr_modem::startRX(int) { _rx_running = true; if (_gr_demod_base) _gr_demod_base->start(); }
void gr_modem::stopRX() { _rx_running = false; if (_gr_demod_base) _gr_demod_base->stop(); }
void gr_modem::startTX(int) { _tx_running = true; }
void gr_modem::stopTX() { _tx_running = false; }
void gr_modem::flushSources() { if (_gr_mod_base) _gr_mod_base->flush_sources(); }
double gr_modem::getFreqGUI() { return _gr_demod_base ? (double)_rx_freq : 0; }
void gr_modem::tune(int64_t center_freq) { if (_gr_demod_base) _rx_freq = center_freq; }
void gr_modem::tuneTx(int64_t center_freq) { if (_gr_mod_base) _tx_freq = center_freq; }   // (band limits are the SDR layer's: src/limits.h)
void gr_modem::setCarrierOffset(int64_t offset) { _rx_offset = offset; if (_gr_demod_base) _gr_demod_base->set_carrier_offset((double)offset); }
void gr_modem::setTxCarrierOffset(int64_t offset) { _tx_offset = offset; if (_gr_mod_base) _gr_mod_base->set_carrier_offset((double)offset); }
qint64 gr_modem::resetTxCarrierOffset()   // gr_mod_base::reset_carrier_offset (src/gr/gr_mod_base.cpp:807-812): back to the offset kept aside, returned
{
    if (!_gr_mod_base) return 0;
    _gr_mod_base->set_carrier_offset((double)_tx_offset);
    return (qint64)_tx_offset;
}
void gr_modem::setSampRate(int samp_rate)
{
    _samp_rate = samp_rate;
    if (_gr_demod_base) _gr_demod_base->set_samp_rate(samp_rate);
    if (_gr_mod_base) _gr_mod_base->set_samp_rate(samp_rate);
}
void gr_modem::setFFTSize(int size) { if (_gr_demod_base) _gr_demod_base->set_fft_size(size); }
void gr_modem::setTxPower(float, std::string) {}                    // the SDR's
void gr_modem::setRxSensitivity(double, std::string) {}            // the SDR's
void gr_modem::setBbGain(int value) { if (_gr_mod_base) _gr_mod_base->set_bb_gain((float)value / 5.0f); }   // :471-476
void gr_modem::setGain(int value) { if (_gr_demod_base) _gr_demod_base->set_gain((float)value / 100.0f); }  // :478-483
void gr_modem::setK(bool value) { if (_gr_mod_base) _gr_mod_base->set_cw_k(value); }
void gr_modem::setAgcAttack(int value) { if (_gr_demod_base) _gr_demod_base->set_agc_attack((float)value); }
void gr_modem::setAgcDecay(int value) { if (_gr_demod_base) _gr_demod_base->set_agc_decay((float)value); }
void gr_modem::setSquelch(int value) { if (_gr_demod_base) _gr_demod_base->set_squelch(value); }
void gr_modem::setFilterWidth(int width)   // :518-524
{
    if (_gr_demod_base) _gr_demod_base->set_filter_width(width, _modem_type_rx);
    if (_gr_mod_base) _gr_mod_base->set_filter_width(width, _modem_type_tx);
}
void gr_modem::setRxCTCSS(float value) { if (_gr_demod_base) _gr_demod_base->set_ctcss(value); }
void gr_modem::setTxCTCSS(float value) { if (_gr_mod_base) _gr_mod_base->set_ctcss(value); }
void gr_modem::enableGUIConst(bool value) { if (_gr_demod_base) _gr_demod_base->enable_gui_const(value); }
void gr_modem::enableGUIFFT(bool value) { if (_gr_demod_base) _gr_demod_base->enable_gui_fft(value); }
void gr_modem::enableTimeDomain(bool value) { if (_gr_demod_base) _gr_demod_base->enable_time_domain(value); }
void gr_modem::enableRSSI(bool value) { if (_gr_demod_base) _gr_demod_base->enable_rssi(value); }
void gr_modem::calibrateRSSI(float value) { if (_gr_demod_base) _gr_demod_base->calibrate_rssi(value); }
void gr_modem::enableDemod(bool value) { if (_gr_demod_base) _gr_demod_base->enable_demodulator(value); }
void gr_modem::getFFTData(float *data, unsigned int &size) { if (_gr_demod_base) _gr_demod_base->get_FFT_data(data, size); }
void gr_modem::getSampleData(float *data, unsigned int &size) { if (_gr_demod_base) _gr_demod_base->get_sample_data(data, size); }
void gr_modem::setSampleWindow(unsigned int size) { if (_gr_demod_base) _gr_demod_base->set_sample_window(size); }
void gr_modem::setTimeDomainSampleRate(unsigned int samp_rate) { if (_gr_demod_base) _gr_demod_base->set_time_sink_samp_rate((int)samp_rate); }
void gr_modem::setTimeDomainFilterWidth(double filter_width) { if (_gr_demod_base) _gr_demod_base->set_time_domain_filter_width(filter_width); }
float gr_modem::getRSSI() { return _gr_demod_base ? _gr_demod_base->get_rssi() : 9999.0f; }   // :604-612
std::vector<gr_complex> *gr_modem::getConstellation() { return _gr_demod_base ? _gr_demod_base->get_constellation_data() : nullptr; }

// ---- TX (:628-978): framing and queueing are gr_modem_hip's (its byte stream is pinned against the reference class: tests/test_gpu_modem_facade.py)
static bool protocol_stack_mode(int m) { return m == 40 /* ModemTypeM17 */ || m == 41 /* ModemTypeDMR */; }
void gr_modem::sendCallsign(QString callsign) { if (_modem && _gr_mod_base) _modem->sendCallsign(callsign.toStdString()); }
void gr_modem::startTransmission(QString callsign)
{
    if (!_gr_mod_base || !_modem) return;
    if (protocol_stack_mode(_modem_type_tx)) {   // M17Transmitter / DMRControl (:684-727): the protocol stacks above this layer
        if (!_warned_protocol && _logger) _logger->log(Logger::LogLevelWarning, QString("HIP gr_modem: the M17 / DMR transmit protocol stacks are not part of this build"));
        _warned_protocol = true;
        return;
    }
    _modem->startTransmission(callsign.toStdString());
}
void gr_modem::endTransmission(QString callsign)
{
    if (!_gr_mod_base || !_modem || protocol_stack_mode(_modem_type_tx)) return;
    _modem->endTransmission(callsign.toStdString());
}
void gr_modem::transmitDMRHeader(unsigned int) {}
void gr_modem::transmitDMR(unsigned char *audio_data, int) { delete[] audio_data; }
void gr_modem::transmitM17Audio(unsigned char *data, int) { delete[] data; }
void gr_modem::transmitTextData(QString text, int frame_type) { if (_modem && _gr_mod_base) _modem->transmitTextData(text.toStdString(), frame_type); }
void gr_modem::transmitBinData(QByteArray bin_data, int frame_type)
{
    if (!_modem || !_gr_mod_base) return;
    const unsigned char *p = reinterpret_cast<const unsigned char *>(bin_data.constData());
    _modem->transmitBinData(std::vector<unsigned char>(p, p + bin_data.size()), frame_type);
}
void gr_modem::transmitDigitalAudio(unsigned char *data, int size) { if (_modem && _gr_mod_base) _modem->transmitDigitalAudio(data, size); else delete[] data; }   // (:812-819: deletes data)
void gr_modem::transmitVideoData(unsigned char *data, int size) { if (_modem && _gr_mod_base) _modem->transmitVideoData(data, size); else delete[] data; }
void gr_modem::transmitNetData(unsigned char *data, int size) { if (_modem && _gr_mod_base) _modem->transmitNetData(data, size); else delete[] data; }
void gr_modem::transmitPCMAudio(std::vector<float> *audio_data)   // :822-831
{
    if (!_gr_mod_base) { audio_data->clear(); delete audio_data; return; }
    _gr_mod_base->set_audio(audio_data);
}

// ---- RX (:996-1117)
bool gr_modem::demodulateAnalog() { return _modem && _gr_demod_base ? _modem->demodulateAnalog() : false; }
bool gr_modem::demodulate() { return _modem && _gr_demod_base ? _modem->demodulate() : false; }

// ---- the SDR side
size_t gr_modem::rxMaxSamples() const { return _rx_max; }
void gr_modem::rxSamples(const gr_complex *iq, size_t n)
{
    if (!_gr_demod_base || !_rx_running) return;
    const gr_complex *in[1] = {iq};
    _gr_demod_base->work(in, n);
}
size_t gr_modem::txMaxSamples() const
{
    if (!_gr_mod_base) return 0;
    return _gr_mod_base->analog() ? _gr_mod_base->max_audio_out() : _gr_mod_base->samples_per_byte() * _tx_max;
}
size_t gr_modem::txSamples(gr_complex *iq, size_t cap)
{
    if (!_gr_mod_base || !_tx_running) return 0;
    if (cap < txMaxSamples()) throw std::invalid_argument("gr_modem::txSamples: the buffer must hold txMaxSamples() samples");
    gr_complex *out[1] = {iq};
    return _gr_mod_base->work(out);
}
