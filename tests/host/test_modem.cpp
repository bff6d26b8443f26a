// Drives the gr_modem-shaped facade (qradiolink_amd/host/gr_modem_hip.*) the way radiocontroller.cpp drives gr_modem:
//   test_modem loopback <modem_type> <streams> <frames> <out.txt>
// Per stream s: startTransmission("CALL<s>"), <frames> voice frames with a known payload, a text message, endTransmission;
// the modulator's samples (+ a little silence and a per-stream delay) go into the demodulator in ragged work() calls;
// demodulate() is polled like the radio loop does.  Every callback is logged to <out.txt> as one line per event.
#include <algorithm>
#include <atomic>
#include <thread>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>

#include "gr_modem_hip.h"

using namespace qrl_host;

static std::string hex(const unsigned char* d, int n)
{
    static const char* h = "0123456789abcdef";
    std::string s;
    for (int i = 0; i < n; ++i) { s.push_back(h[d[i] >> 4]); s.push_back(h[d[i] & 15]); }
    return s;
}

int main(int argc, char** argv)
{
    if (argc == 4 && !strcmp(argv[1], "txpin")) {
        // test_modem txpin <modem_type> <out.bin>: a fixed script of gr_modem's TX entry points; the bytes the facade hands to the byte
        // source (gr_mod_base::set_data) are written out and compared with what the REFERENCE's gr_modem produces for the same script
        // (tests/test_gpu_modem_facade.py, oracle/ref_shim_modem.cpp)
        const int mode = atoi(argv[2]);
        try {
            qrl_runtime rt(0);
            struct tap_mod : gr_mod_base_hip {
                using gr_mod_base_hip::gr_mod_base_hip;
                std::vector<uint8_t> sent;
                int set_data(std::vector<uint8_t>* data, int stream) override { if (stream == 0) sent.insert(sent.end(), data->begin(), data->end()); delete data; return 1; }
            } mod(rt, 1, 1000000, 0.0, 4096);
            gr_modem_events ev;
            gr_modem_hip modem(nullptr, &mod, ev);
            modem.toggleTxMode(mode);
            const int L = modem_tx_frame_length(mode);
            modem.startTransmission("N0CALL");
            for (int f = 0; f < 3; ++f) {
                unsigned char* d = new unsigned char[L];
                for (int i = 0; i < L; ++i) d[i] = (unsigned char)(31 * f + 7 * i + 1);
                modem.transmitDigitalAudio(d, L);
            }
            modem.transmitTextData("the quick brown fox jumps over the lazy dog 0123456789");
            std::vector<unsigned char> bin(2 * L + 3);
            for (size_t i = 0; i < bin.size(); ++i) bin[i] = (unsigned char)(200 - i);
            modem.transmitBinData(bin);
            { unsigned char* d = new unsigned char[L]; for (int i = 0; i < L; ++i) d[i] = (unsigned char)(i ^ 0x5A); modem.transmitVideoData(d, L); }
            { unsigned char* d = new unsigned char[L]; for (int i = 0; i < L; ++i) d[i] = (unsigned char)(i * 3); modem.transmitNetData(d, L); }
            modem.sendCallsign("AB1CD");
            modem.endTransmission("N0CALL");
            std::ofstream o(argv[3], std::ios::binary);
            o.write(reinterpret_cast<const char*>(mod.sent.data()), (std::streamsize)mod.sent.size());
            std::printf("txpin ok: %zu bytes\n", mod.sent.size());
            return 0;
        } catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
    }
    if (argc == 3 && !strcmp(argv[1], "robust")) {
        // test_modem robust <dmr iq out.bin>: a rejected setting must not cost the caller the handle it had, and the TX setters serialise with work()
        try {
            qrl_runtime rt(0);
            auto expect_throw = [](const char* what, auto&& fn) {
                try { fn(); } catch (const std::exception& e) { std::printf("  %s: rejected (%s)\n", what, e.what()); return; }
                throw std::runtime_error(std::string(what) + ": expected an exception");
            };
            {   // TX: set_carrier_offset on modes without the back end keeps the modulator and its offset
                for (int mode : {QRL_MODEM_M17, QRL_MODEM_BPSK8}) {
                    const size_t maxb = mode == QRL_MODEM_M17 ? 96 : 3;   // (DSSS makes 1 000 000 samples per byte)
                    gr_mod_base_hip m2(rt, 1, 1000000, 0.0, maxb);
                    std::vector<gr_complex> b2(mode == QRL_MODEM_M17 ? 96 / 3 * 2500 + 2500 : 4 * 1000000); gr_complex* p2[1] = {b2.data()};
                    m2.set_mode(mode);
                    expect_throw("tx offset without back end", [&] { m2.set_carrier_offset(12500.0); });
                    m2.set_data(new std::vector<uint8_t>(maxb == 96 ? 48 : 3, 0x5A));
                    if (m2.work(p2) == 0) throw std::runtime_error("modulator lost after a rejected set_carrier_offset");
                    m2.set_mode(mode);   // and set_mode still works (the offset did not stick)
                }
                gr_mod_base_hip mod(rt, 1, 1000000, 0.0, 96);
                std::vector<gr_complex> buf(96 * 2500 + 2500); gr_complex* ptr[1] = {buf.data()};
                // a mode WITH the back end: the first non-zero offset re-opens the handle with the rotator; queued bytes of the old handle are dropped
                mod.set_mode(QRL_MODEM_GMSK10K);
                mod.set_data(new std::vector<uint8_t>(10, 0x33));
                mod.set_carrier_offset(5000.0);
                if (mod.work(ptr) != 0) throw std::runtime_error("bytes queued for the old handle survived the re-open");
                mod.set_data(new std::vector<uint8_t>(10, 0x33));
                if (mod.work(ptr) != 10 * mod.samples_per_byte()) throw std::runtime_error("modulator with back end produced nothing");
                mod.set_carrier_offset(0.0);   // has a back end now: plain retune
                expect_throw("tx unknown mode", [&] { mod.set_mode(9999); });
                mod.set_data(new std::vector<uint8_t>(10, 0x33));
                if (mod.work(ptr) == 0) throw std::runtime_error("modulator lost after a rejected set_mode");
            }
            {   // TX: a facade built for fewer than 1024 items per call keys CW
                gr_mod_base_hip mod(rt, 1, 1000000, 0.0, 1023);
                mod.set_mode(QRL_MODEM_CW600USB);
                mod.set_cw_k(true);
                std::vector<gr_complex> buf(mod.max_audio_out() + 16); gr_complex* ptr[1] = {buf.data()};
                mod.work(ptr); mod.work(ptr);
                if (mod.cw_samples_per_call() != 1023) throw std::runtime_error("cw_samples_per_call not clamped to max");
            }
            {   // TX: setDMRData from another thread while work() runs -- every frame's idle zeros must be applied (frame + 39 zero bytes = 72 bytes = 60000 samples,
                // of which the zero run silences 780 x 20... checked as: the number of EXACT zero samples equals frames x run length)
                gr_mod_base_hip mod(rt, 1, 1000000, 0.0, 144);
                mod.set_mode(QRL_MODEM_DMR);
                std::vector<gr_complex> buf(144 / 3 * 2500 + 2500), all; gr_complex* ptr[1] = {buf.data()};
                const int frames = 24;
                std::atomic<bool> done{false};
                std::thread feeder([&] {
                    for (int f = 0; f < frames; ++f) {
                        std::vector<uint8_t> fr(33);
                        for (int i = 0; i < 33; ++i) fr[i] = (uint8_t)(0x1B + 7 * i + f);
                        mod.setDMRData({fr}, 0);
                        std::this_thread::sleep_for(std::chrono::microseconds(200 * (f % 5)));
                    }
                    done = true;
                });
                for (;;) {
                    const bool fin = done.load();
                    const size_t got = mod.work(ptr);
                    all.insert(all.end(), buf.begin(), buf.begin() + got);
                    if (fin && got == 0) break;
                }
                feeder.join();
                // one stream, whole 3-byte blocks per pass: whatever the interleaving of the two threads, the byte stream is frame | 39 zeros | frame ... and
                // every zero run must be known to the pass that reaches its bytes -- the IQ is compared with the oracle's gr_mod_dmr by the Python test
                std::printf("  dmr concurrent: %zu samples from %d frames\n", all.size(), frames);
                if (all.size() != (size_t)frames * 72 / 3 * 2500) throw std::runtime_error("DMR sample count");
                std::ofstream o(argv[2], std::ios::binary);
                o.write(reinterpret_cast<const char*>(all.data()), (std::streamsize)(all.size() * sizeof(gr_complex)));
            }
            // RX: scope settings the engine rejects restore the previous ones; the demodulator keeps running
            gr_demod_base_hip demod(rt, 1, 1000000, 0.0, 1 << 15);
            expect_throw("rx scope rate without a handle", [&] { demod.set_time_sink_samp_rate(800000); });   // no handle yet: range-checked
            demod.set_mode(QRL_MODEM_2FSK1K);
            demod.enable_time_domain(true);
            demod.set_time_sink_samp_rate(50000);
            int thrown = 0;
            try { demod.set_time_sink_samp_rate(1000); } catch (const std::invalid_argument&) { ++thrown; }        // low_pass of > 4096 taps
            try { demod.set_time_domain_filter_width(100.0); } catch (const std::invalid_argument&) { ++thrown; }   // same
            try { demod.set_time_sink_samp_rate(700000); } catch (const std::invalid_argument&) { ++thrown; }      // above the engine's range
            if (thrown != 3) throw std::runtime_error("rejected scope settings did not throw");
            std::vector<gr_complex> x(1 << 15, gr_complex(0.01f, 0.0f));
            const gr_complex* in[1] = {x.data()};
            demod.work(in, x.size());
            demod.work(in, x.size());
            demod.flush();
            std::vector<float> buf(2 * 8096 + 2); unsigned ns = 0; size_t items = 0;
            for (;;) { demod.get_sample_data(buf.data(), ns, 0); if (!ns) break; items += ns / 2; }
            std::printf("  rx scope after rejected settings: %zu items at the restored 50 ksps\n", items);
            if (items < 2 * x.size() / 20 - 600 || items > 2 * x.size() / 20) throw std::runtime_error("scope tap not running at the restored rate");
            demod.set_mode(QRL_MODEM_GMSK10K);   // later mode changes still work
            std::printf("robust ok\n");
            return 0;
        } catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
    }
    if (argc == 4 && !strcmp(argv[1], "dmrtx")) {
        // test_modem dmrtx <frames.bin: 5 x 33 bytes> <iq prefix>: gr_mod_base::setDMRData on the TX facade, two radios -- stream 0 sends frames 0..2 (the
        // third one queued after the first work()), stream 1 frames 3..4
        try {
            qrl_runtime rt(0);
            gr_mod_base_hip mod(rt, 2, 1000000, 0.0, 144);
            mod.set_mode(QRL_MODEM_DMR);
            std::ifstream f(argv[2], std::ios::binary);
            std::vector<uint8_t> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
            if (raw.size() != 5 * 33) throw std::runtime_error("frames.bin: 165 bytes expected");
            auto fr = [&](int i) { return std::vector<uint8_t>(raw.begin() + 33 * i, raw.begin() + 33 * (i + 1)); };
            std::vector<std::vector<gr_complex>> iq(2), buf(2, std::vector<gr_complex>(144 / 3 * 2500 + 2500));
            std::vector<gr_complex*> ptr{buf[0].data(), buf[1].data()};
            auto run = [&]() {
                size_t got;
                while ((got = mod.work(ptr.data())) != 0)
                    for (int s = 0; s < 2; ++s) iq[s].insert(iq[s].end(), buf[s].begin(), buf[s].begin() + got);
            };
            mod.setDMRData({fr(0), fr(1)}, 0);
            mod.setDMRData({fr(3)}, 1);
            run();
            mod.setDMRData({fr(2)}, 0);
            mod.setDMRData({fr(4)}, 1);
            run();
            for (int s = 0; s < 2; ++s) {
                std::ofstream o(std::string(argv[3]) + std::to_string(s) + ".bin", std::ios::binary);
                o.write(reinterpret_cast<const char*>(iq[s].data()), (std::streamsize)(iq[s].size() * sizeof(gr_complex)));
            }
            std::printf("dmrtx ok\n");
            return 0;
        } catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
    }
    if (argc == 6 && !strcmp(argv[1], "scoperate")) {
        // test_modem scoperate <iq.bin: [1][n] cf32 at 1 Msps> <out.bin> <time sink samp rate or 0> <time domain filter width or 0>: the facade's scope tap after
        // set_time_sink_samp_rate / set_time_domain_filter_width, drained through get_sample_data; the complex items go to out.bin
        try {
            qrl_runtime rt(0);
            gr_demod_base_hip demod(rt, 1, 1000000, 0.0, 1 << 16);
            demod.set_mode(QRL_MODEM_2FSK1K);
            demod.enable_time_domain(true);
            demod.set_sample_window(8096);
            if (atoi(argv[4])) demod.set_time_sink_samp_rate(atoi(argv[4]));
            if (atof(argv[5]) > 0) demod.set_time_domain_filter_width(atof(argv[5]));
            demod.set_time_sink_samp_rate(2000000);   // > 1 Msps: ignored (gr_demod_base.cpp:1251-1252)
            std::ifstream f(argv[2], std::ios::binary);
            std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
            const size_t n = raw.size() / sizeof(gr_complex);
            const gr_complex* x = reinterpret_cast<const gr_complex*>(raw.data());
            std::vector<gr_complex> items;
            std::vector<float> buf(2 * 8096 + 2);
            auto drain = [&]() {
                for (;;) {
                    unsigned ns = 0;
                    demod.get_sample_data(buf.data(), ns, 0);
                    if (!ns) break;
                    for (unsigned i = 0; i < ns / 2; ++i) items.emplace_back(buf[i], buf[ns / 2 + i + 1]);   // reals, then imaginaries from index n + 1 on
                }
            };
            for (size_t pos = 0; pos < n; pos += 50000) {
                const size_t take = std::min<size_t>(n - pos, 50000) & ~(size_t)1;
                if (!take) break;
                const gr_complex* ptr = x + pos;
                demod.work(&ptr, take);
                drain();
            }
            demod.flush();
            drain();
            std::ofstream o(argv[3], std::ios::binary);
            o.write(reinterpret_cast<const char*>(items.data()), (std::streamsize)(items.size() * sizeof(gr_complex)));
            std::printf("scoperate ok: %zu items\n", items.size());
            return 0;
        } catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
    }
    if (argc == 4 && !strcmp(argv[1], "cw")) {
        // test_modem cw <streams> <iq prefix>: the CW branch of the TX facade: key up for 3 calls of 1024 tone samples, down for 4 (the key set in another
        // mode: kept), up again for 3
        const int N = atoi(argv[2]);
        try {
            qrl_runtime rt(0);
            gr_mod_base_hip mod(rt, N, 1000000, 0.0, 4096);
            mod.set_mode(QRL_MODEM_CW600USB);
            mod.set_cw_samples_per_call(1024);
            std::vector<std::vector<gr_complex>> iq(N), buf(N, std::vector<gr_complex>(mod.max_audio_out()));
            std::vector<gr_complex*> ptr(N);
            for (int s = 0; s < N; ++s) ptr[s] = buf[s].data();
            auto run = [&](int calls) {
                for (int c = 0; c < calls; ++c) {
                    const size_t got = mod.work(ptr.data());
                    for (int s = 0; s < N; ++s) iq[s].insert(iq[s].end(), buf[s].begin(), buf[s].begin() + got);
                }
            };
            run(3);
            mod.set_cw_k(true);
            run(4);
            mod.set_cw_k(false);
            run(3);
            for (int s = 0; s < N; ++s) {
                std::ofstream o(std::string(argv[3]) + std::to_string(s) + ".bin", std::ios::binary);
                o.write(reinterpret_cast<const char*>(iq[s].data()), (std::streamsize)(iq[s].size() * sizeof(gr_complex)));
            }
            std::printf("cw ok\n");
            return 0;
        } catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
    }
    if ((argc == 8 || argc == 10) && !strcmp(argv[1], "analogtx")) {
        // test_modem analogtx <modem_type> <streams> <audio.bin: [streams][n] f32> <iq prefix> <set_filter_width value or 0> <ctcss tone or 0> [<device rate> <offset Hz>]:
        // the TX facade's analogue path -- set_mode, the setters (before AND, for the width, once more after a detour through another mode: the
        // reference's instances keep them), set_audio in ragged pieces, work() until the queues are empty
        const int mode = atoi(argv[2]), N = atoi(argv[3]), width = atoi(argv[6]);
        const float tone = (float)atof(argv[7]);
        try {
            qrl_runtime rt(0);
            const int rate = argc == 10 ? atoi(argv[8]) : 1000000;
            const double offset = argc == 10 ? atof(argv[9]) : 0.0;
            gr_mod_base_hip mod(rt, N, rate, offset, 4096);
            if (width) mod.set_filter_width(width, mode);            // before the mode exists: remembered for its instance
            if (tone != 0.0f) mod.set_ctcss(tone);
            mod.set_mode(QRL_MODEM_QPSK2K);                          // a detour through a digital mode
            mod.set_mode(mode);
            if (!mod.analog()) throw std::runtime_error("not an analogue mode");
            mod.set_bb_gain(0.75f);
            std::ifstream f(argv[4], std::ios::binary);
            std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
            const size_t n = raw.size() / sizeof(float) / (size_t)N;
            const float* a = reinterpret_cast<const float*>(raw.data());
            std::vector<std::vector<gr_complex>> iq(N);
            std::vector<std::vector<gr_complex>> buf(N, std::vector<gr_complex>(mod.max_audio_out()));
            if (mod.samples_per_audio_sample() != (size_t)125 * (size_t)(rate / 1000000)) throw std::runtime_error("samples_per_audio_sample");
            std::vector<gr_complex*> ptr(N);
            for (int s = 0; s < N; ++s) ptr[s] = buf[s].data();
            static const size_t sizes[] = {640, 4096, 3, 1000, 2049};
            size_t pos = 0; unsigned k = 0;
            while (pos < n) {
                const size_t take = std::min(n - pos, sizes[k++ % 5]);
                for (int s = 0; s < N; ++s) mod.set_audio(new std::vector<float>(a + (size_t)s * n + pos, a + (size_t)s * n + pos + take), s);
                pos += take;
                size_t got;
                while ((got = mod.work(ptr.data())) != 0)
                    for (int s = 0; s < N; ++s) iq[s].insert(iq[s].end(), buf[s].begin(), buf[s].begin() + got);
            }
            for (int s = 0; s < N; ++s) {
                std::ofstream o(std::string(argv[5]) + std::to_string(s) + ".bin", std::ios::binary);
                o.write(reinterpret_cast<const char*>(iq[s].data()), (std::streamsize)(iq[s].size() * sizeof(gr_complex)));
            }
            // set_carrier_offset on a handle without the back end: zero is a no-op, a non-zero offset re-opens the handle with the rotator
            if (rate == 1000000 && offset == 0.0) {
                mod.set_carrier_offset(0.0);
                mod.set_carrier_offset(12500.0);
                for (int s = 0; s < N; ++s) mod.set_audio(new std::vector<float>(2048 + 4, 0.25f), s);
                if (mod.work(ptr.data()) == 0) throw std::runtime_error("no output after the retune");
                mod.set_carrier_offset(-2500.0);                         // now a phase-continuous retune of the open handle
            }
            std::printf("analogtx ok\n");
            return 0;
        } catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
    }
    if ((argc == 6 || argc == 8) && !strcmp(argv[1], "analog")) {
        // test_modem analog <modem_type> <streams> <iq.bin: [streams][n] cf32> <audio prefix>: the facade's analogue path the way
        // radiocontroller polls it (gr_modem::demodulateAnalog -> pcmAudio); stream s's audio goes to <prefix><s>.bin
        const int mode = atoi(argv[2]), N = atoi(argv[3]);
        try {
            qrl_runtime rt(0);
            std::vector<std::vector<float>> audio(N);
            gr_modem_events ev;
            ev.pcmAudio = [&](int s, std::vector<float>* pcm) { audio[s].insert(audio[s].end(), pcm->begin(), pcm->end()); delete pcm; };
            const size_t chunk = 1 << 16;
            gr_demod_base_hip demod(rt, N, 1000000, 0.0, chunk);
            gr_modem_hip modem(&demod, nullptr, ev);
            if (argc == 8) {   // ... <set_filter_width value or 0> <set_gain value in thousandths or 0>: before the mode exists, remembered for its instance
                if (atoi(argv[6])) demod.set_filter_width(atoi(argv[6]), mode);
                if (atoi(argv[7])) demod.set_gain((float)atoi(argv[7]) / 1000.0f);
                demod.set_filter_width(1234, QRL_MODEM_QPSK2K);   // the reference's default branch: ignored
            }
            modem.toggleRxMode(mode);
            std::ifstream f(argv[4], std::ios::binary);
            std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
            const size_t n = raw.size() / sizeof(gr_complex) / (size_t)N;
            const gr_complex* x = reinterpret_cast<const gr_complex*>(raw.data());
            static const size_t sizes[] = {65536, 4096, 2, 33334, 20000};
            size_t pos = 0; unsigned k = 0;
            while (pos < n) {
                const size_t take = std::min(n - pos, sizes[k++ % 5]) & ~(size_t)1;
                if (!take) break;
                std::vector<const gr_complex*> ptr(N);
                for (int s = 0; s < N; ++s) ptr[s] = x + (size_t)s * n + pos;
                demod.work(ptr.data(), take);
                pos += take;
                for (int s = 0; s < N; ++s) while (modem.demodulateAnalog(s)) {}      // 640-sample packets, like radiocontroller's poll
            }
            demod.flush();
            for (int s = 0; s < N; ++s) while (modem.demodulateAnalog(s)) {}
            for (int s = 0; s < N; ++s) {
                std::ofstream o(std::string(argv[5]) + std::to_string(s) + ".bin", std::ios::binary);
                o.write(reinterpret_cast<const char*>(audio[s].data()), (std::streamsize)(audio[s].size() * sizeof(float)));
            }
            std::printf("analog ok\n");
            return 0;
        } catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
    }
    if (argc == 5 && !strcmp(argv[1], "hosttime")) {
        // test_modem hosttime <modem_type> <streams> <calls>: host CPU time of the RX boundary per work() call, with the reference's
        // per-bit loop on the host (set_device_framing(false)) and with the frame synchroniser on the device (the default): every
        // stream carries the same modulated frames; prints the time spent in work() and in the demodulate() polls and the factor
        const int mode = atoi(argv[2]), N = atoi(argv[3]), calls = atoi(argv[4]);
        try {
            qrl_runtime rt(0);
            const size_t chunk = 1 << 14;
            std::vector<gr_complex> txs;
            {
                gr_mod_base_hip mod(rt, 1, 1000000, 0.0, 4096);
                gr_modem_events none;
                gr_modem_hip txm(nullptr, &mod, none);
                txm.toggleTxMode(mode);
                const int L = modem_tx_frame_length(mode);
                txm.startTransmission("N0CALL");
                while (txs.size() < chunk * (size_t)calls) {
                    for (int f = 0; f < 8; ++f) {
                        unsigned char* d = new unsigned char[L];
                        for (int i = 0; i < L; ++i) d[i] = (unsigned char)(31 * f + 7 * i + 1);
                        if (mode == QRL_MODEM_QPSK250K || mode == QRL_MODEM_QPSKVIDEO || mode == QRL_MODEM_4FSK100K) txm.transmitNetData(d, L);   // the fast modes frame IP / video only (gr_modem.cpp:1183-1282)
                        else txm.transmitDigitalAudio(d, L);
                    }
                    std::vector<gr_complex> part(mod.samples_per_byte() * 4096);
                    for (;;) { gr_complex* o = part.data(); const size_t ns = mod.work(&o); if (!ns) break; for (size_t i = 0; i < ns; ++i) txs.push_back(0.05f * part[i]); }
                }
            }
            double res[2][3] = {{0, 0, 0}, {0, 0, 0}};
            for (int dev = 0; dev < 2; ++dev) {
                long frames = 0;
                gr_modem_events ev;
                ev.digitalAudio = [&](int, const unsigned char*, int) { ++frames; };
                ev.netData = [&](int, const unsigned char*, int) { ++frames; };
                gr_demod_base_hip demod(rt, N, 1000000, 0.0, chunk);
                gr_modem_hip modem(&demod, nullptr, ev);
                modem.set_device_framing(dev == 1);
                modem.toggleRxMode(mode);
                std::vector<const gr_complex*> in(N);
                double t_work = 0, t_poll = 0;
                for (int k = 0; k < calls; ++k) {
                    for (int s = 0; s < N; ++s) in[s] = txs.data() + (size_t)k * chunk;
                    const auto t0 = std::chrono::steady_clock::now();
                    demod.work(in.data(), chunk);
                    const auto t1 = std::chrono::steady_clock::now();
                    for (int s = 0; s < N; ++s) while (modem.demodulate(s)) {}
                    const auto t2 = std::chrono::steady_clock::now();
                    if (k >= 2) { t_work += std::chrono::duration<double, std::milli>(t1 - t0).count(); t_poll += std::chrono::duration<double, std::milli>(t2 - t1).count(); }
                }
                demod.flush();
                for (int s = 0; s < N; ++s) while (modem.demodulate(s)) {}
                res[dev][0] = t_work / std::max(1, calls - 2); res[dev][1] = t_poll / std::max(1, calls - 2); res[dev][2] = (double)frames;
            }
            std::printf("hosttime mode %d streams %d calls %d chunk %zu: host loop work %.3f ms + poll %.3f ms per call (%.0f frames); device framing work %.3f ms + poll %.3f ms per call (%.0f frames); "
                        "poll factor %.1f, whole boundary factor %.2f\n", mode, N, calls, chunk, res[0][0], res[0][1], res[0][2], res[1][0], res[1][1], res[1][2],
                        res[0][1] / std::max(res[1][1], 1e-6), (res[0][0] + res[0][1]) / std::max(res[1][0] + res[1][1], 1e-6));
            return res[0][2] == res[1][2] && res[0][2] > 0 ? 0 : 3;
        } catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
    }
    if (argc == 3 && !strcmp(argv[1], "valveoff")) {
        // test_modem valveoff <streams>: enable_demodulator(false) + enable_gui_fft(true) -- only the spectrum tap listens (gr_demod_base.cpp:1150-1153).
        // Every call carries a tone at a DIFFERENT frequency; the spectrum polled after a call must peak at THAT call's tone (a spectrum of stale
        // or half-uploaded samples peaks at the previous call's).  Prints one line per call: expected bin, peak bin.
        const int N = atoi(argv[2]);
        try {
            qrl_runtime rt(0);
            const size_t chunk = 1 << 15;
            gr_demod_base_hip demod(rt, N, 1000000, 0.0, chunk);
            demod.set_mode(QRL_MODEM_2FSK1K);
            demod.set_fft_size(4096);
            demod.enable_gui_fft(true);
            demod.enable_demodulator(false);
            std::vector<std::vector<gr_complex>> x(N, std::vector<gr_complex>(chunk));
            std::vector<float> spectrum(4096);
            int bad = 0;
            for (int k = 0; k < 12; ++k) {
                const int bin = 300 + 290 * k;                       // cycles per 4096 samples
                for (int s = 0; s < N; ++s)
                    for (size_t i = 0; i < chunk; ++i) { const double ph = 2 * M_PI * (double)((long)((bin + s) * i) % 4096) / 4096.0; x[s][i] = gr_complex(0.1f * (float)std::cos(ph), 0.1f * (float)std::sin(ph)); }
                std::vector<const gr_complex*> in(N);
                for (int s = 0; s < N; ++s) in[s] = x[s].data();
                demod.work(in.data(), chunk);
                for (int s = 0; s < N; ++s) {
                    unsigned got = 4096;
                    if (s == 0) demod.get_FFT_data(spectrum.data(), got, 0);          // fetches the frame of every stream ...
                    else if (const float* p = demod.last_FFT_data(s)) std::copy(p, p + 4096, spectrum.begin());   // ... the others read it from there
                    else got = 0;
                    if (got != 4096) { std::printf("call %d stream %d: no spectrum\n", k, s); ++bad; continue; }
                    int pk = 0; for (int i = 1; i < 4096; ++i) if (spectrum[i] > spectrum[pk]) pk = i;
                    const int want = (bin + s + 2048) % 4096;        // half swap: DC in the middle
                    std::printf("call %d stream %d: want %d peak %d (%.1f dB)\n", k, s, want, pk, spectrum[pk]);
                    if (std::abs(pk - want) > 1) ++bad;
                }
            }
            return bad ? 3 : 0;
        } catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
    }
    if (argc != 6 || strcmp(argv[1], "loopback")) { std::fprintf(stderr, "usage: test_modem loopback modem_type streams frames out.txt\n"); return 2; }
    const int mode = atoi(argv[2]), N = atoi(argv[3]), nframes = atoi(argv[4]);
    try {
        qrl_runtime rt(0);
        std::ofstream log(argv[5]);
        gr_modem_events ev;
        ev.digitalAudio = [&](int s, const unsigned char* d, int n) { log << s << " audio " << hex(d, n) << "\n"; };
        ev.textReceived = [&](int s, const std::string& t, bool) { log << s << " text " << hex(reinterpret_cast<const unsigned char*>(t.data()), (int)t.size()) << "\n"; };
        ev.callsignReceived = [&](int s, const std::string& c) { log << s << " callsign " << c << "\n"; };
        ev.dataFrameReceived = [&](int s) { log << s << " dataframe\n"; };
        ev.endAudioTransmission = [&](int s) { log << s << " endaudio\n"; };
        ev.receiveEnd = [&](int s) { log << s << " receiveend\n"; };
        // QRL_TEST_TAP=<file>: every bit vector demodulate() pulls out of the mailboxes ("B stream nr bits"), every demodulate() call
        // ("D stream") and every event ("E stream text") in order: tests/test_gpu_modem_facade.py replays the bits into the REFERENCE's
        // gr_modem and compares the events
        FILE* tap = getenv("QRL_TEST_TAP") ? std::fopen(getenv("QRL_TEST_TAP"), "w") : nullptr;
        if (tap) {
            ev.digitalAudio = [&, tap](int s, const unsigned char* d, int n) { log << s << " audio " << hex(d, n) << "\n"; std::fprintf(tap, "E %d audio %s\n", s, hex(d, n).c_str()); };
            ev.textReceived = [&, tap](int s, const std::string& t, bool h) { log << s << " text " << hex(reinterpret_cast<const unsigned char*>(t.data()), (int)t.size()) << "\n";
                                                                          std::fprintf(tap, "E %d %s %s\n", s, h ? "html" : "text", hex(reinterpret_cast<const unsigned char*>(t.data()), (int)t.size()).c_str()); };
            ev.callsignReceived = [&, tap](int s, const std::string& c) { log << s << " callsign " << c << "\n"; std::fprintf(tap, "E %d callsign %s\n", s, c.c_str()); };
            ev.dataFrameReceived = [&, tap](int s) { log << s << " dataframe\n"; std::fprintf(tap, "E %d dataframe\n", s); };
            ev.endAudioTransmission = [&, tap](int s) { log << s << " endaudio\n"; std::fprintf(tap, "E %d endaudio\n", s); };
            ev.receiveEnd = [&, tap](int s) { log << s << " receiveend\n"; std::fprintf(tap, "E %d receiveend\n", s); };
            ev.videoData = [tap](int s, const unsigned char* d, int n) { std::fprintf(tap, "E %d video %s\n", s, hex(d, n).c_str()); };
            ev.netData = [tap](int s, const unsigned char* d, int n) { std::fprintf(tap, "E %d net %s\n", s, hex(d, n).c_str()); };
            ev.protoReceived = [tap](int s, const std::vector<unsigned char>& d) { std::fprintf(tap, "E %d proto %s\n", s, hex(d.data(), (int)d.size()).c_str()); };
        }
        struct tap_demod : gr_demod_base_hip {
            using gr_demod_base_hip::gr_demod_base_hip;
            FILE* tap = nullptr;
            std::vector<unsigned char>* getData(int nr, int stream) override
            {
                std::vector<unsigned char>* v = gr_demod_base_hip::getData(nr, stream);
                if (tap && v) { std::string b(v->size(), '0'); for (size_t i = 0; i < v->size(); ++i) b[i] = (char)('0' + ((*v)[i] & 1)); std::fprintf(tap, "B %d %d %s\n", stream, nr, b.c_str()); }
                return v;
            }
        };
        const size_t chunk = 1 << 17;
        tap_demod demod(rt, N, 1000000, 0.0, chunk);
        demod.tap = tap;
        gr_mod_base_hip mod(rt, N, 1000000, 0.0, 4096);
        gr_modem_hip modem(&demod, &mod, ev);
        if (getenv("QRL_TEST_BRANCH")) modem.set_branch_rule(gr_modem_hip::BranchRuleReference);   // the reference's literal `>=` rule (like-for-like replay)
        // default: the frame synchroniser runs on the device (qrl_framesync_* behind the demodulator).  QRL_TEST_HOSTLOOP=1 keeps the
        // reference's per-bit loop on the host -- the checker.  With the tap on, the raw bit ports are still copied out so that the
        // python side can replay them into the reference's gr_modem (the device path itself never looks at them).
        const bool host_loop = getenv("QRL_TEST_HOSTLOOP") != nullptr;
        modem.set_device_framing(!host_loop);
        if (tap && !host_loop) demod.keep_bits(true);
        const bool two = modem_two_branches(mode);
        auto poll = [&](int s) {
            for (;;) {
                if (tap) std::fprintf(tap, "D %d\n", s);
                if (tap && !host_loop) { delete demod.getData(1, s); if (two) delete demod.getData(2, s); }   // logs the "B" lines of this poll
                const bool active = modem.demodulate(s);
                if (tap) std::fprintf(tap, "R %d %d\n", s, active ? 1 : 0);   // the return value radiocontroller.cpp:1298 polls
                if (!active) break;
            }
        };
        modem.toggleTxMode(mode);
        modem.toggleRxMode(mode);
        demod.enable_rssi(true);
        demod.calibrate_rssi(-30.0f);
        demod.set_fft_size(4096);
        demod.enable_gui_fft(true);
        demod.enable_time_domain(true);           // the scope tap (gr_demod_base::enable_time_domain)
        demod.set_sample_window(4001);            // odd: the sink makes it 4002
        std::vector<float> scope(2 * 4002 + 2);
        size_t scope_items = 0, scope_reads = 0; double scope_power = 0.0; unsigned scope_max = 0;
        std::vector<float> spectrum(4096);
        int spectra = 0; float peak_db = -1000.0f; int peak_bin = -1;
        std::vector<float> rssi_max(N, -1000.0f);
        const int L = modem_tx_frame_length(mode);
        for (int s = 0; s < N; ++s) {
            modem.startTransmission("CALL" + std::to_string(s), s);
            for (int f = 0; f < nframes; ++f) {
                unsigned char* d = new unsigned char[L];
                for (int i = 0; i < L; ++i) d[i] = (unsigned char)(17 * s + 31 * f + 7 * i + 1);
                modem.transmitDigitalAudio(d, L, s);
            }
            modem.transmitTextData("hello from stream " + std::to_string(s), (int)FrameTypeText, s);
            modem.endTransmission("CALL" + std::to_string(s), s);
        }
        // modulate everything that is queued
        const size_t spb = mod.samples_per_byte();
        std::vector<std::vector<gr_complex>> tx(N);
        std::vector<std::vector<gr_complex>> part(N, std::vector<gr_complex>(spb * 4096));
        for (;;) {
            std::vector<gr_complex*> outp(N);
            for (int s = 0; s < N; ++s) outp[s] = part[s].data();
            const size_t ns = mod.work(outp.data());
            if (!ns) break;
            for (int s = 0; s < N; ++s) tx[s].insert(tx[s].end(), part[s].begin(), part[s].begin() + ns);
        }
        // channel: per-stream delay, scale, trailing silence so that the last frames leave the filters / Viterbi
        size_t total = 0;
        for (int s = 0; s < N; ++s) total = std::max(total, tx[s].size());
        total += 40000 + 1000 * N;
        total &= ~(size_t)1;
        std::vector<std::vector<gr_complex>> rx(N, std::vector<gr_complex>(total, gr_complex(0, 0)));
        for (int s = 0; s < N; ++s)
            for (size_t i = 0; i < tx[s].size(); ++i) rx[s][i + 500 * s] = 0.05f * tx[s][i];   // (SDR-like level: the reference's FLL loop gain scales with the input power, there is no AGC in front of it)
        // feed in ragged, even-sized calls and poll like the radio loop (radiocontroller.cpp:1291-1303)
        size_t pos = 0;
        size_t sizes[] = {65536, 10000, 131072, 2, 77778};
        if (const char* e = getenv("QRL_TEST_CHUNK")) for (auto& v : sizes) v = (size_t)atol(e);
        FILE* bitlog = getenv("QRL_TEST_BITS") ? std::fopen(getenv("QRL_TEST_BITS"), "wb") : nullptr;
        for (int k = 0; pos < total; ++k) {
            const size_t n = std::min(sizes[k % 5], total - pos) & ~(size_t)1;
            if (!n) break;
            std::vector<const gr_complex*> in(N);
            for (int s = 0; s < N; ++s) in[s] = rx[s].data() + pos;
            demod.work(in.data(), n);
            pos += n;
            if (getenv("QRL_TEST_DEBUG")) {
                for (int s = 0; s < N; ++s) {
                    auto* a = demod.getData(1, s); auto* b = demod.getData(2, s);
                    std::fprintf(stderr, "call %d stream %d: bits A %zu, bits B %zu, tx %zu samples\n", k, s, a ? a->size() : 0, b ? b->size() : 0, tx[s].size());
                    if (bitlog && s == 0 && a) std::fwrite(a->data(), 1, a->size(), bitlog);
                    delete a; delete b;
                }
                continue;
            }
            for (int s = 0; s < N; ++s) poll(s);
            for (int s = 0; s < N; ++s) { const float v = demod.get_rssi(s); if (v != 0.0f) rssi_max[s] = std::max(rssi_max[s], v); }   // (0 = the probe before its first item)
            for (;;) {                                      // the GUI timer drains the sample sink window by window
                unsigned ns = 0;
                demod.get_sample_data(scope.data(), ns, 0);
                if (!ns) break;
                ++scope_reads; scope_items += ns / 2; scope_max = std::max(scope_max, ns);
                for (unsigned i = 0; i < ns / 2; ++i) scope_power += (double)scope[i] * scope[i] + (double)scope[ns / 2 + i + 1] * scope[ns / 2 + i + 1];
            }
            unsigned got = 0;
            demod.get_FFT_data(spectrum.data(), got, 0);   // the GUI timer of the reference polls like this
            if (got == 4096) {
                ++spectra;
                for (int i = 0; i < 4096; ++i) if (spectrum[i] > peak_db) { peak_db = spectrum[i]; peak_bin = i; }
            }
        }
        if (bitlog) std::fclose(bitlog);
        demod.flush();
        for (int s = 0; s < N; ++s) poll(s);
        if (tap) std::fclose(tap);
        log << "0 framing " << (modem.device_framing() ? "device" : "host") << "\n";
        for (int s = 0; s < N; ++s) log << s << " modem_sync " << modem.modem_sync(s) << "\n";
        for (int s = 0; s < N; ++s) log << s << " rssi " << demod.get_rssi(s) << "\n" << s << " rssi_max " << rssi_max[s] << "\n";
        log << "0 spectra " << spectra << "\n" << "0 peak_bin " << peak_bin << "\n" << "0 peak_db " << peak_db << "\n";
        log << "0 scope_items " << scope_items << "\n" << "0 scope_reads " << scope_reads << "\n" << "0 scope_max " << scope_max << "\n"
            << "0 scope_power " << (scope_items ? scope_power / (double)scope_items : 0.0) << "\n" << "0 samples_in " << total << "\n";
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "exception: %s\n", e.what());
        return 1;
    }
}
