// gr_golden.cpp — runs ONE reference demodulator hier block on a cf32 file with the real GNU Radio 3.10 runtime and dumps
// its ports.  Not built here (no GNU Radio in this container); see README.md.  Links against the reference's own sources
// for the hier blocks (-DQRADIOLINK_SRC=...), contains none of them.
//   gr_golden <2fsk|gmsk|qpsk|4fsk|bpsk|dmr> <sps> <filter_width> <fm> <in.cf32> <out_prefix>
#include <gnuradio/blocks/file_sink.h>
#include <gnuradio/blocks/file_source.h>
#include <gnuradio/blocks/head.h>
#include <gnuradio/top_block.h>
#include <cstdlib>
#include <iostream>
#include <string>
#include "gr/gr_demod_2fsk.h"
#include "gr/gr_demod_4fsk.h"
#include "gr/gr_demod_bpsk.h"
#include "gr/gr_demod_dmr.h"
#include "gr/gr_demod_gmsk.h"
#include "gr/gr_demod_qpsk.h"

int main(int argc, char** argv)
{
    if (argc != 7) { std::cerr << "usage: gr_golden family sps filter_width fm in.cf32 out_prefix\n"; return 2; }
    const std::string fam = argv[1], in = argv[5], out = argv[6];
    const int sps = std::atoi(argv[2]), fw = std::atoi(argv[3]), fm = std::atoi(argv[4]);
    gr::top_block_sptr tb = gr::make_top_block("gr_golden");
    gr::basic_block_sptr demod;
    int nports = 4;
    if (fam == "2fsk") demod = make_gr_demod_2fsk(sps, 1000000, 1700, fw, fm != 0);
    else if (fam == "gmsk") demod = make_gr_demod_gmsk(sps, 1000000, 1700, fw);
    else if (fam == "bpsk") demod = make_gr_demod_bpsk(sps, 1000000, 1700, fw);
    else if (fam == "qpsk") { demod = make_gr_demod_qpsk(sps, 1000000, 1700, fw); nports = 3; }
    else if (fam == "4fsk") { demod = make_gr_demod_4fsk(sps, 1000000, 1700, fw, fm != 0); nports = 3; }
    else if (fam == "dmr") { demod = make_gr_demod_dmr(sps, 1000000); nports = 4; }
    else return 2;
    auto src = gr::blocks::file_source::make(sizeof(gr_complex), in.c_str(), false);
    tb->connect(src, 0, demod, 0);
    // ports: 0 filtered (cf32), 1 constellation (cf32), 2 bits A (u8), 3 bits B (u8; float for dmr)
    const size_t item[4] = {sizeof(gr_complex), sizeof(gr_complex), 1, fam == "dmr" ? sizeof(float) : 1};
    for (int p = 0; p < nports; ++p) {
        auto sink = gr::blocks::file_sink::make(item[p], (out + ".port" + std::to_string(p)).c_str());
        sink->set_unbuffered(false);
        tb->connect(demod, p, sink, 0);
    }
    tb->run();   // file_source without repeat: the flowgraph drains and stops
    return 0;
}
