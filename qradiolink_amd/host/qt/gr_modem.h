// gr_modem.h -- a `class gr_modem` with the reference's public interface (/root/reference/src/gr_modem.h:55-139: the same methods,
// signals and slots, the same argument types and defaults) over the HIP path, so that radiocontroller.cpp (call sites
// src/radiocontroller.cpp:121-152 connects, :1298 demodulate(), :1302 demodulateAnalog(), :1969-2078 init / tune / start) compiles against
// it unchanged.  Replaces src/gr_modem.h + src/gr_modem.cpp + src/gr/gr_demod_base.* + src/gr/gr_mod_base.* in a QRadioLink build
// (drop this directory in front of src/ on the include path and link libqrl_host.a + libqrl_hip.so).
//
// Built against real Qt in a maintainer's build (moc generates the signals); in this repository's tests against oracle/qt_stub -- the
// same sliver of Qt the reference's own gr_modem.cpp is compiled with for the pins -- and driven by the SAME driver source as the
// reference class (tests/host/gr_modem_script.h), signal log against signal log (tests/test_gpu_modem_literal.py).
//
// What is NOT behind this boundary: the SDR.  The reference's gr_demod_base / gr_mod_base own an osmosdr source /
// sink inside their flow graphs; here the device-rate IQ crosses the boundary explicitly through the two methods at the end of the public
// section (rxSamples / txSamples) -- the one addition to the reference's interface.  The hardware controls (tune, gains, antenna,
// frequency correction) keep their signatures and store what they are given.  The M17 and DMR protocol stacks (M17Transmitter,
// DMRControl: src/M17, src/DMR) stay what they are in the reference: transmitM17Audio / transmitDMR / transmitDMRHeader and the M17 / DMR
// branches of start / endTransmission are declared and do nothing but release their buffers (logged once).
#ifndef GR_MODEM_H
#define GR_MODEM_H

#include <QObject>
#include <QString>
#include <QVector>
#include <QByteArray>
#include <QMap>
#include <complex>
#include <memory>
#include <string>
#include <vector>
#include "src/settings.h"
#include "src/logger.h"
#include "src/layer1framing.h"

class DMRControl;
typedef std::complex<float> gr_complex;
class qrl_runtime;
namespace qrl_host { class gr_demod_base_hip; class gr_mod_base_hip; class gr_modem_hip; }

class gr_modem : public QObject
{
    Q_OBJECT
public:
    explicit gr_modem(const Settings *settings, Logger *logger, DMRControl *dmrcontrol, QObject *parent = 0);
    ~gr_modem();

    bool demodulateAnalog();
    void sendCallsign(QString callsign);

signals:
    void pcmAudio(std::vector<float>* pcm);
    void digitalAudio(unsigned char *c2data, int size);
    void videoData(unsigned char *video_data, int size);
    void netData(unsigned char *net_data, int size);
    void demodulated_audio(short *pcm, short size);
    void textReceived(QString text, bool html);
    void protoReceived(QByteArray data);
    void callsignReceived(QString text);
    void m17FrameInfoReceived(QString src, QString dest, uint16_t CAN);
    void audioFrameReceived();
    void dataFrameReceived();
    void syncIssues();
    void receiveEnd();
    void endAudioTransmission();
    void endBeep();

public slots:
    void transmitPCMAudio(std::vector<float> *audio_data);
    void transmitDigitalAudio(unsigned char *data, int size);
    void transmitM17Audio(unsigned char *data, int size);
    void transmitVideoData(unsigned char *data, int size);
    void transmitNetData(unsigned char *data, int size);
    void transmitDMR(unsigned char *audio_data, int size);
    void transmitDMRHeader(unsigned int ts);
    bool demodulate();
    void startTransmission(QString callsign);
    void endTransmission(QString callsign);
    void transmitTextData(QString text, int frame_type = FrameTypeText);
    void transmitBinData(QByteArray bin_data, int frame_type = FrameTypeProto);
    void initTX(int modem_type, int64_t frequency, std::string device_args,
                std::string device_antenna, int freq_corr, int initial_gain=94, int mmdvm_channels=3,
                int mmdvm_channel_separation=25000);
    void initRX(int modem_type, std::string device_args,
                std::string device_antenna, int freq_corr, int mmdvm_channels=3,
                int mmdvm_channel_separation=25000);
    void deinitTX(int modem_type);
    void deinitRX(int modem_type);
    void toggleRxMode(int modem_type);
    void toggleTxMode(int modem_type);
    void tune(int64_t center_freq);
    void tuneTx(int64_t center_freq);
    void startRX(int buffer_size=0);
    void stopRX();
    void startTX(int buffer_size=0);
    void stopTX();
    void setTxPower(float value, std::string gain_stage="");
    void setBbGain(int value);
    void setGain(int value);
    void setK(bool value);
    void setSquelch(int value);
    void setFilterWidth(int filter_width);
    void setRxSensitivity(double value, std::string gain_stage="");
    void setAgcAttack(int value);
    void setAgcDecay(int value);
    void setRxCTCSS(float value);
    void setTxCTCSS(float value);
    void enableGUIConst(bool value);
    void enableGUIFFT(bool value);
    void enableTimeDomain(bool value);
    void enableRSSI(bool value);
    void calibrateRSSI(float value);
    void enableDemod(bool value);
    double getFreqGUI();
    void getFFTData(float *data, unsigned int &size);
    void getSampleData(float *data, unsigned int &size);
    void setSampleWindow(unsigned int size);
    void setTimeDomainSampleRate(unsigned int samp_rate);
    void setTimeDomainFilterWidth(double filter_width);
    void setCarrierOffset(int64_t offset);
    void setTxCarrierOffset(int64_t offset);
    qint64 resetTxCarrierOffset();
    void setSampRate(int samp_rate);
    void setFFTSize(int size);
    float getRSSI();
    void flushSources();
    std::vector<gr_complex> *getConstellation();
    const QMap<std::string, QVector<int> > getRxGainNames() const;
    const QMap<std::string, QVector<int> > getTxGainNames() const;

public:
    // ---- where the SDR was (the only additions to the reference's interface) ----
    // rxSamples: n (even, <= rxMaxSamples()) new device-rate samples from the SDR driver into the receive path -- what the osmosdr source inside
    // gr_demod_base's flow graph delivered to the scheduler (src/gr/gr_demod_base.cpp:150-200).  Ignored between stopRX() and startRX().
    void rxSamples(const gr_complex *iq, size_t n);
    size_t rxMaxSamples() const;
    // txSamples: one scheduler pass of the transmit path: what is queued becomes at most `cap` device-rate samples in `iq` for the SDR sink
    // (src/gr/gr_mod_base.cpp:249-262); returns the number written (0: nothing queued, or between stopTX() and startTX()).
    size_t txSamples(gr_complex *iq, size_t cap);
    size_t txMaxSamples() const;
    // the device (HIP ordinal) this modem runs on and the batch geometry; call before initRX / initTX.  Defaults: device 0, 65536 samples per
    // rxSamples call, 4096 bytes per transmit pass.
    void setDevice(int device, size_t rx_max_samples = 65536, size_t tx_max_bytes = 4096);
    // two-branch modes (both Viterbi alignments decoded, src/gr_modem.cpp:1048-1090): false (default) = each branch keeps its own frame synchroniser and
    // frames of either alignment are delivered; true = the reference's literal rule -- the longer bit vector wins, `>=` favours branch 1 -- which with the
    // device's equal counts means branch 1 only (the like-for-like setting of tests/test_gpu_modem_literal.py).  Call before initRX.
    void setReferenceBranchRule(bool value) { _reference_branch_rule = value; }

protected:
    // factory hooks (tests derive a class that taps the bit mailboxes; a maintainer has no reason to touch them)
    virtual qrl_host::gr_demod_base_hip *createDemodBase(qrl_runtime &rt, int samp_rate, double offset, size_t max_samples);
    virtual qrl_host::gr_mod_base_hip *createModBase(qrl_runtime &rt, int samp_rate, double offset, size_t max_bytes);

private:
    void rebuildModem();
    const Settings *_settings;
    Logger *_logger;
    DMRControl *_dmr_control;
    std::unique_ptr<qrl_runtime> _rt;
    qrl_host::gr_demod_base_hip *_gr_demod_base;
    qrl_host::gr_mod_base_hip *_gr_mod_base;
    qrl_host::gr_modem_hip *_modem;
    int _modem_type_rx, _modem_type_tx;
    int _device; size_t _rx_max, _tx_max;
    int _samp_rate;
    int64_t _rx_freq, _tx_freq, _rx_offset, _tx_offset;
    bool _rx_running, _tx_running, _warned_protocol, _reference_branch_rule = false;
    std::vector<gr_complex> _tx_buf;
};

#endif // GR_MODEM_H
