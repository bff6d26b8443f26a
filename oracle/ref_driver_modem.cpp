// ref_driver_modem.cpp -- TEST INFRASTRUCTURE.  The reference's OWN gr_modem class (/root/reference/src/gr_modem.cpp compiled where it lies, against
// oracle/qt_stub) driven by the SAME driver source as the HIP path's class of that name (tests/host/gr_modem_script.h):
//     gr_modem_script_ref <mode> <frames> <tap.txt> <log.txt> <tx.bin>
// runs the TX script (the bytes its gr_mod_base stub receives go to tx.bin), then replays the bit vectors of tap.txt -- what the HIP demodulator handed
// to ITS gr_modem, poll by poll -- into the stub gr_demod_base and polls demodulate() at the same points.  log.txt must equal the HIP side's log.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
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
#include "gr_modem_script.h"

int main(int argc, char** argv)
{
    if (argc != 6) { std::fprintf(stderr, "usage: gr_modem_script_ref <mode> <frames> <tap.txt> <log.txt> <tx.bin>\n"); return 2; }
    const int mode = std::atoi(argv[1]), nframes = std::atoi(argv[2]);
    g_script_log = std::fopen(argv[4], "w");
    if (!g_script_log) return 1;
    Settings settings; Logger logger; DMRControl dmr;
    gr_modem m(&settings, &logger, &dmr);
    script_setup(m, mode);
    script_transmit(m, mode, nframes, "YO8RZZ");
    {
        const std::vector<unsigned char>& s = m._gr_mod_base->sent;
        std::ofstream o(argv[5], std::ios::binary);
        o.write(reinterpret_cast<const char*>(s.data()), (std::streamsize)s.size());
    }
    std::ifstream tap(argv[3]);
    std::string line;
    bool pending = false;
    while (std::getline(tap, line)) {
        if (line == "D") {
            if (pending) script_poll(m);
            pending = true;
        } else if (line.size() > 4 && line[0] == 'B') {
            const int nr = line[2] - '0';
            std::vector<unsigned char>* v = new std::vector<unsigned char>(line.size() - 4);
            for (size_t i = 4; i < line.size(); ++i) (*v)[i - 4] = (unsigned char)(line[i] - '0');
            // two-branch modes poll getData(1) and getData(2); the others getData() = queue 0 (src/gr_modem.cpp:1048-1071)
            const bool two = m._modem_type_rx == 0 || (m._modem_type_rx >= 15 && m._modem_type_rx <= 22) || m._modem_type_rx == 24 || m._modem_type_rx == 25;
            m._gr_demod_base->q[two ? nr : 0].push_back(v);
        }
    }
    if (pending) script_poll(m);
    std::fclose(g_script_log);
    return 0;
}
