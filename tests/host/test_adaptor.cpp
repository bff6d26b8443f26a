// Drives the GNU Radio-shaped adaptor blocks the way the scheduler would: work() with arbitrary item counts.
//   test_adaptor nodevice                      -> expects std::runtime_error from construction (exit 0 if thrown)
//   test_adaptor rx <family>[+d<type>] <sps> <fw> <fm> <device_rate> <offset> <iq.bin> <bits_a.bin> <bits_b.bin>
//                (family 2fsk | gmsk | qpsk | 4fsk | bpsk | dmr; "+d2" attaches gr_deframer_bb(2) to ports 2/3)
//   test_adaptor rxa <nbfm | am | wbfm | usb | lsb> <fw> <iq.bin> <audio.bin>      analogue receivers: port 1 = audio mailbox
//   test_adaptor m17seq <frames.bin: [n][48]> <out.bin>           M17FrameDecoder-shaped host class: n frames through ONE radio's
//                decoder; out = n type bytes + the 30 LSF bytes + the 18 stream-frame bytes it holds afterwards
//   test_adaptor txa <fw> <audio.bin: f32> <iq.bin>                 make_gr_mod_nbfm-shaped block, ragged work() calls
//   test_adaptor tx <bytes.bin> <iq.bin> [family sps fw fm]
#include "gr_hip_blocks.h"
#include "m17_frame_decoder_hip.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>

static std::vector<char> slurp(const char* path)
{
    std::ifstream f(path, std::ios::binary);
    return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static void dump(const char* path, const void* p, size_t n)
{
    std::ofstream f(path, std::ios::binary);
    f.write(static_cast<const char*>(p), (std::streamsize)n);
}

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    try {
        if (!strcmp(argv[1], "nodevice")) {
            try { qrl_runtime rt(0); } catch (const std::runtime_error& e) { std::printf("runtime_error: %s\n", e.what()); return 0; }
            std::printf("a device is present\n");
            return 0;
        }
        qrl_runtime rt(0);
        if (!strcmp(argv[1], "rx") && argc == 11) {
            std::string fam = argv[2];
            int deframer = 0;
            const size_t plus = fam.find("+d");
            if (plus != std::string::npos) { deframer = atoi(fam.c_str() + plus + 2); fam = fam.substr(0, plus); }
            const int sps = atoi(argv[3]), fw = atoi(argv[4]), fm = atoi(argv[5]), rate = atoi(argv[6]);
            gr_demod_hip_sptr d = fam == "2fsk" ? make_gr_demod_2fsk_hip(rt, sps, 1000000, 1700, fw, fm != 0)
                                : fam == "gmsk" ? make_gr_demod_gmsk_hip(rt, sps, 1000000, 1700, fw)
                                : fam == "4fsk" ? make_gr_demod_4fsk_hip(rt, sps, 1000000, 1700, fw, fm != 0)
                                : fam == "bpsk" ? make_gr_demod_bpsk_hip(rt, sps, 1000000, 1700, fw)
                                : fam == "dmr"  ? make_gr_demod_dmr_hip(rt, sps, 1000000)
                                                : make_gr_demod_qpsk_hip(rt, sps, 1000000, 1700, fw);
            if (deframer) d->attach_deframer(deframer);
            if (rate != 1000000) d->set_device_samp_rate(rate);
            d->set_carrier_offset(atof(argv[7]));
            std::vector<char> raw = slurp(argv[8]);
            const gr_complex* x = reinterpret_cast<const gr_complex*>(raw.data());
            const size_t n = raw.size() / sizeof(gr_complex);
            std::vector<unsigned char> a, b;
            size_t pos = 0; unsigned k = 0;
            static const int sizes[] = {8191, 4096, 1, 33333, 7, 65536, 12345};
            gr_vector_void_star outs;
            while (pos < n) {
                const size_t take = std::min<size_t>(n - pos, (size_t)sizes[k++ % 7]);
                gr_vector_const_void_star ins(1, x + pos);
                if (d->work((int)take, ins, outs) != (int)take) return 3;
                pos += take;
                for (int nr = 1; nr <= 2; ++nr)
                    if (std::vector<unsigned char>* v = d->get_data(nr)) { (nr == 1 ? a : b).insert((nr == 1 ? a : b).end(), v->begin(), v->end()); delete v; }
            }
            dump(argv[9], a.data(), a.size());
            dump(argv[10], b.data(), b.size());
            std::printf("rx ok: %zu samples -> %zu / %zu bits\n", n, a.size(), b.size());
            return 0;
        }
        if (!strcmp(argv[1], "m17seq") && argc == 4) {
            std::vector<char> raw = slurp(argv[2]);
            const size_t n = raw.size() / 48;
            qrl_host::m17_frame_decoder_hip dec(rt, 2, 4);     // (capacity 4: the batch is cut into several device calls)
            std::vector<int> where(n, 1);
            const auto types = dec.decodeFrames(reinterpret_cast<const uint8_t*>(raw.data()), where.data(), n);
            std::vector<uint8_t> out;
            for (auto t : types) out.push_back(static_cast<uint8_t>(t));
            out.insert(out.end(), dec.getLsf(1).begin(), dec.getLsf(1).end());
            out.insert(out.end(), dec.getStreamFrame(1).begin(), dec.getStreamFrame(1).end());
            out.insert(out.end(), dec.getLsf(0).begin(), dec.getLsf(0).end());       // the other radio's state stays clear
            dump(argv[3], out.data(), out.size());
            std::printf("m17seq ok: %zu frames\n", n);
            return 0;
        }
        if (!strcmp(argv[1], "txa") && argc == 5) {
            gr_amod_hip_sptr m = make_gr_mod_nbfm_hip(rt, 20, 1000000, 1700, atoi(argv[2]));
            m->set_bb_gain(0.75f);
            std::vector<char> raw = slurp(argv[3]);
            const float* a = reinterpret_cast<const float*>(raw.data());
            const size_t n = raw.size() / sizeof(float);
            std::vector<gr_complex> iq, buf;
            static const size_t sizes[] = {320, 1, 1023, 6, 2000, 5};
            size_t pos = 0; unsigned k = 0;
            while (pos < n) {
                const size_t take = std::min(n - pos, sizes[k++ % 6]);
                buf.resize((take + 3) * 125);
                gr_vector_const_void_star ins(1, a + pos);
                gr_vector_void_star outs(1, buf.data());
                const int r = m->work((int)(take * 125), ins, outs);
                iq.insert(iq.end(), buf.begin(), buf.begin() + r);
                pos += take;
            }
            dump(argv[4], iq.data(), iq.size() * sizeof(gr_complex));
            std::printf("txa ok: %zu audio samples -> %zu IQ samples\n", n, iq.size());
            return 0;
        }
        if (!strcmp(argv[1], "rxa") && argc == 6) {
            const std::string fam = argv[2];
            const int fw = atoi(argv[3]);
            gr_demod_hip_sptr d = fam == "nbfm" ? make_gr_demod_nbfm_hip(rt, 125, 1000000, 1700, fw)
                                : fam == "am"   ? make_gr_demod_am_hip(rt, 125, 1000000, 1700, fw)
                                : fam == "usb"  ? make_gr_demod_ssb_hip(rt, 125, 1000000, 1700, fw, 0)
                                : fam == "lsb"  ? make_gr_demod_ssb_hip(rt, 125, 1000000, 1700, fw, 1)
                                                : make_gr_demod_wbfm_hip(rt, 125, 1000000, 1700, fw);
            std::vector<char> raw = slurp(argv[4]);
            const gr_complex* x = reinterpret_cast<const gr_complex*>(raw.data());
            const size_t n = raw.size() / sizeof(gr_complex);
            std::vector<float> audio;
            size_t pos = 0; unsigned k = 0;
            static const int sizes[] = {8191, 4096, 1, 33333, 7, 65536, 12345};
            gr_vector_void_star outs;
            while (pos < n) {
                const size_t take = std::min<size_t>(n - pos, (size_t)sizes[k++ % 7]);
                gr_vector_const_void_star ins(1, x + pos);
                if (d->work((int)take, ins, outs) != (int)take) return 3;
                pos += take;
                while (std::vector<float>* v = d->get_audio_data()) { audio.insert(audio.end(), v->begin(), v->end()); delete v; }   // 640-sample packets
            }
            dump(argv[5], audio.data(), audio.size() * sizeof(float));
            std::printf("rxa ok: %zu samples -> %zu audio samples\n", n, audio.size());
            return 0;
        }
        if (!strcmp(argv[1], "tx") && (argc == 4 || argc == 8)) {
            const std::string fam = argc == 8 ? argv[4] : "qpsk";
            const int sps = argc == 8 ? atoi(argv[5]) : 4, fw = argc == 8 ? atoi(argv[6]) : 160000, fm = argc == 8 ? atoi(argv[7]) : 0;
            gr_mod_hip_sptr m = fam == "2fsk" ? make_gr_mod_2fsk_hip(rt, sps, 1000000, 1700, fw, fm != 0)
                              : fam == "gmsk" ? make_gr_mod_gmsk_hip(rt, sps, 1000000, 1700, fw)
                              : fam == "4fsk" ? make_gr_mod_4fsk_hip(rt, sps, 1000000, 1700, fw, fm != 0)
                              : fam == "bpsk" ? make_gr_mod_bpsk_hip(rt, sps, 1000000, 1700, fw)
                                              : make_gr_mod_qpsk_hip(rt, sps, 1000000, 1700, fw);
            const size_t I = m->interpolation();
            std::vector<char> raw = slurp(argv[2]);
            std::vector<gr_complex> out(raw.size() * I);
            size_t pos = 0; unsigned k = 0;
            static const int nb[] = {1, 100, 9000, 17, 2048};
            while (pos < raw.size()) {
                const size_t take = std::min<size_t>(raw.size() - pos, (size_t)nb[k++ % 5]);
                gr_vector_const_void_star ins(1, raw.data() + pos);
                gr_vector_void_star outs(1, out.data() + pos * I);
                if (m->work((int)(take * I), ins, outs) != (int)(take * I)) return 3;
                pos += take;
            }
            dump(argv[3], out.data(), out.size() * sizeof(gr_complex));
            std::printf("tx ok: %zu bytes -> %zu samples\n", raw.size(), out.size());
            return 0;
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "exception: %s\n", e.what());
        return 1;
    }
    return 2;
}
