// test_gr_modem_literal -- the HIP path's `class gr_modem` (qradiolink_amd/host/qt/gr_modem.*, built against oracle/qt_stub in place of Qt) driven by
// tests/host/gr_modem_script.h:   test_gr_modem_literal <mode> <frames> <log.txt> <tap.txt> <tx.bin>
// TX script -> txSamples() -> a loop-back channel (level, delay; the whole transmission twice, the second time one channel bit later, so that in the
// two-branch modes both Viterbi alignments occur) -> rxSamples() in ragged calls -> demodulate() polled like radiocontroller.cpp:1291-1303.
// log.txt: "S ..." per signal, "R 0|1" per demodulate() call.  tap.txt: "D" per poll and "B <nr> <bits>" per bit vector the poll consumed -- what
// oracle/_ref/gr_modem_script_ref replays into the reference's own class.  tx.bin: the bytes handed to the modulator's byte source.
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <stdexcept>

#include "qt/gr_modem.h"
#include "gr_modem_hip.h"
#include "gr_modem_script.h"

using namespace qrl_host;

static FILE* g_tap = nullptr;
static std::vector<uint8_t> g_sent;
struct tap_demod : gr_demod_base_hip {
    using gr_demod_base_hip::gr_demod_base_hip;
    void log_vectors()   // the vectors this poll's demodulate() accounts for (kept beside the device synchroniser: keep_bits)
    {
        for (int nr = 1; nr <= 2; ++nr) {
            std::vector<unsigned char>* v = gr_demod_base_hip::getData(nr, 0);
            if (!v) continue;
            std::string b(v->size(), '0');
            for (size_t i = 0; i < v->size(); ++i) b[i] = (char)('0' + ((*v)[i] & 1));
            std::fprintf(g_tap, "B %d %s\n", nr, b.c_str());
            delete v;
        }
    }
};
struct tap_mod : gr_mod_base_hip {
    using gr_mod_base_hip::gr_mod_base_hip;
    int set_data(std::vector<uint8_t>* data, int stream) override { g_sent.insert(g_sent.end(), data->begin(), data->end()); return gr_mod_base_hip::set_data(data, stream); }
};
struct tap_modem : gr_modem {
    using gr_modem::gr_modem;
    tap_demod* demod = nullptr;
    gr_demod_base_hip* createDemodBase(qrl_runtime& rt, int rate, double off, size_t n) override { demod = new tap_demod(rt, 1, rate, off, n); demod->keep_bits(true); return demod; }
    gr_mod_base_hip* createModBase(qrl_runtime& rt, int rate, double off, size_t n) override { return new tap_mod(rt, 1, rate, off, n); }
};

int main(int argc, char** argv)
{
    if (argc != 6) { std::fprintf(stderr, "usage: test_gr_modem_literal <mode> <frames> <log.txt> <tap.txt> <tx.bin>\n"); return 2; }
    const int mode = std::atoi(argv[1]), nframes = std::atoi(argv[2]);
    try {
        g_script_log = std::fopen(argv[3], "w");
        g_tap = std::fopen(argv[4], "w");
        if (!g_script_log || !g_tap) throw std::runtime_error("cannot open the logs");
        Settings settings; Logger logger;
        tap_modem m(&settings, &logger, nullptr);
        m.setDevice(0, 65536, 4096);
        m.setReferenceBranchRule(std::getenv("QRL_TEST_BOTH_BRANCHES") == nullptr);
        script_setup(m, mode);
        script_transmit(m, mode, nframes, "YO8RZZ");
        std::vector<gr_complex> tx, part(m.txMaxSamples());
        for (;;) { const size_t n = m.txSamples(part.data(), part.size()); if (!n) break; tx.insert(tx.end(), part.begin(), part.begin() + n); }
        { std::ofstream o(argv[5], std::ios::binary); o.write(reinterpret_cast<const char*>(g_sent.data()), (std::streamsize)g_sent.size()); }
        // channel: SDR-like level, 500 samples of delay; the transmission again behind 30000 samples of silence, one channel bit (samples per byte / 16) later
        const size_t bit = std::max<size_t>(1, tx.size() / std::max<size_t>(g_sent.size(), 1) / 16);
        const size_t second = 500 + tx.size() + 30000 + bit + (bit & 1);
        std::vector<gr_complex> rx((second + tx.size() + 60000) & ~(size_t)1, gr_complex(0, 0));
        for (size_t i = 0; i < tx.size(); ++i) { rx[500 + i] = 0.05f * tx[i]; rx[second + i] = 0.05f * tx[i]; }
        auto poll = [&] {
            for (;;) {
                std::fprintf(g_tap, "D\n");
                m.demod->log_vectors();
                if (!script_poll(m)) break;
            }
        };
        const size_t sizes[] = {65536, 10000, 32768, 2, 50002};
        size_t pos = 0;
        for (int k = 0; pos < rx.size(); ++k) {
            const size_t n = std::min(sizes[k % 5], rx.size() - pos);
            m.rxSamples(rx.data() + pos, n);
            pos += n;
            poll();
        }
        m.stopRX();      // waits for the call in flight and harvests it
        poll();
        std::fclose(g_script_log); std::fclose(g_tap);
        g_script_log = nullptr;
        std::printf("literal ok: %zu tx bytes, %zu tx samples\n", g_sent.size(), tx.size());
        return 0;
    } catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
}
