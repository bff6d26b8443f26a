// oracle/ref_shim_rec.cpp — TEST INFRASTRUCTURE, never linked into the product.
//
// The reference's hier-block constructors (src/gr/gr_demod_*.cpp, gr_mod_*.cpp — compiled UNMODIFIED from /root/reference by
// `make -C oracle rec`) run here against oracle/rec_stub/, a stand-in for the stock GNU Radio headers whose ::make() factories,
// setters, firdes designers and hier_block2::connect() only RECORD their arguments.  rr_construct() builds one block and returns
// the construction log: the list of stock blocks with the exact parameters the reference computed for them, the firdes calls
// behind every tap vector, and the wiring.  tests/test_ref_chains.py compares that log with the chain description this repository
// implements (oracle/orc_chains.c / engine.cpp), which pins the PARAMETERS AND WIRING of those chains to the reference's own code.
// The stock blocks' ARITHMETIC stays [GR-MEM] (GNU Radio itself is not in the image).
#include <cstring>
#include <string>
#include <time.h>

#include "src/gr/gr_demod_2fsk.h"
#include "src/gr/gr_demod_am.h"
#include "src/gr/gr_demod_bpsk.h"
#include "src/gr/gr_demod_gmsk.h"
#include "src/gr/gr_demod_m17.h"
#include "src/gr/gr_demod_nbfm.h"
#include "src/gr/gr_demod_qpsk.h"
#include "src/gr/gr_demod_wbfm.h"
#include "src/gr/gr_mod_2fsk.h"
#include "src/gr/gr_mod_4fsk.h"
#include "src/gr/gr_mod_am.h"
#include "src/gr/gr_mod_bpsk.h"
#include "src/gr/gr_demod_4fsk.h"
#include "src/gr/gr_demod_dmr.h"
#include "src/gr/gr_demod_dsss.h"
#include "src/gr/gr_demod_mmdvm.h"
#include "src/gr/gr_demod_mmdvm_multi.h"
#include "src/gr/gr_demod_mmdvm_multi2.h"
#include "src/gr/gr_demod_ssb.h"
#include "src/gr/gr_mod_dmr.h"
#include "src/gr/gr_mod_dsss.h"
#include "src/gr/gr_mod_mmdvm.h"
#include "src/gr/gr_mod_mmdvm_multi.h"
#include "src/gr/gr_mod_mmdvm_multi2.h"
#include "src/gr/gr_mod_ssb.h"
#include "src/gr/gr_mod_gmsk.h"
#include "src/gr/gr_mod_m17.h"
#include "src/gr/gr_mod_nbfm.h"
#include "src/gr/gr_mod_qpsk.h"

static std::string g_text;

// the build renames nanosleep (bursttimer.cpp / gr_mmdvm_source.cpp sleep in their work loops; nothing here runs them)
extern "C" int qrl_stub_nanosleep(const struct timespec*, struct timespec*) { return 0; }

extern "C" {

// Builds reference hier block `kind` with the given constructor arguments and returns its construction log (one event per line);
// nullptr for an unknown kind.  The returned pointer is valid until the next call.
const char* rr_construct(const char* kind, int sps, int samp_rate, int carrier_freq, int filter_width, int fm)
{
    gr::rec::State& s = gr::rec::st();
    s.lines.clear();
    s.designs.clear();
    s.next = 1;
    const std::string k(kind);
    bool ok = true;
    if (k == "demod_2fsk") { auto p = make_gr_demod_2fsk(sps, samp_rate, carrier_freq, filter_width, fm != 0); }
    else if (k == "demod_gmsk") { auto p = make_gr_demod_gmsk(sps, samp_rate, carrier_freq, filter_width); }
    else if (k == "demod_qpsk") { auto p = make_gr_demod_qpsk(sps, samp_rate, carrier_freq, filter_width); }
    else if (k == "demod_bpsk") { auto p = make_gr_demod_bpsk(sps, samp_rate, carrier_freq, filter_width); }
    else if (k == "demod_m17") { auto p = make_gr_demod_m17(sps, samp_rate, carrier_freq, filter_width); }
    else if (k == "demod_nbfm") { auto p = make_gr_demod_nbfm(sps, samp_rate, carrier_freq, filter_width); }
    else if (k == "demod_am") { auto p = make_gr_demod_am(sps, samp_rate, carrier_freq, filter_width); }
    else if (k == "demod_wbfm") { auto p = make_gr_demod_wbfm(sps, samp_rate, carrier_freq, filter_width); }
    else if (k == "mod_2fsk") { auto p = make_gr_mod_2fsk(sps, samp_rate, carrier_freq, filter_width, fm != 0); }
    else if (k == "mod_4fsk") { auto p = make_gr_mod_4fsk(sps, samp_rate, carrier_freq, filter_width, fm != 0); }
    else if (k == "mod_gmsk") { auto p = make_gr_mod_gmsk(sps, samp_rate, carrier_freq, filter_width); }
    else if (k == "mod_qpsk") { auto p = make_gr_mod_qpsk(sps, samp_rate, carrier_freq, filter_width); }
    else if (k == "mod_m17") { auto p = make_gr_mod_m17(sps, samp_rate, carrier_freq, filter_width); }
    else if (k == "mod_nbfm") { auto p = make_gr_mod_nbfm(sps, samp_rate, carrier_freq, filter_width); }
    else if (k == "mod_bpsk") { auto p = make_gr_mod_bpsk(sps, samp_rate, carrier_freq, filter_width); }
    else if (k == "demod_4fsk") { auto p = make_gr_demod_4fsk(sps, samp_rate, carrier_freq, filter_width, fm != 0); }
    else if (k == "demod_dmr") { auto p = make_gr_demod_dmr(sps, samp_rate); }
    else if (k == "demod_dsss") { auto p = make_gr_demod_dsss(sps, samp_rate, carrier_freq, filter_width); }
    else if (k == "demod_ssb") { auto p = make_gr_demod_ssb(sps, samp_rate, carrier_freq, filter_width, fm); }
    else if (k == "demod_mmdvm") { auto p = make_gr_demod_mmdvm(); }
    else if (k == "mod_mmdvm") { auto p = make_gr_mod_mmdvm(); }
    else if (k == "mod_dmr") { auto p = make_gr_mod_dmr(); }
    else if (k == "mod_dsss") { auto p = make_gr_mod_dsss(sps, samp_rate, carrier_freq, filter_width); }
    else if (k == "mod_ssb") { auto p = make_gr_mod_ssb(sps, samp_rate, carrier_freq, filter_width, fm); }
    // the multi-carrier MMDVM graphs: (num_channels, channel_separation, use_tdma) travel in the first three argument slots
    else if (k == "demod_mmdvm_multi") { static BurstTimer bt; auto p = make_gr_demod_mmdvm_multi(&bt, sps, samp_rate, carrier_freq != 0); }
    else if (k == "demod_mmdvm_multi2") { static BurstTimer bt; auto p = make_gr_demod_mmdvm_multi2(&bt, sps, samp_rate, carrier_freq != 0); }
    else if (k == "mod_mmdvm_multi") { static BurstTimer bt; auto p = make_gr_mod_mmdvm_multi(&bt, sps, samp_rate, carrier_freq != 0); }
    else if (k == "mod_mmdvm_multi2") { static BurstTimer bt; auto p = make_gr_mod_mmdvm_multi2(&bt, sps, samp_rate, carrier_freq != 0); }
    else if (k == "mod_am") { auto p = make_gr_mod_am(sps, samp_rate, carrier_freq, filter_width); }
    // "<kind>.set_filter_width" / ".set_gain": construct with the instance's arguments of gr_demod_base.cpp / gr_mod_base.cpp, then call the setter the
    // facade forwards (gr_demod_base::set_filter_width / set_gain, gr_mod_base::set_filter_width) with the value in the LAST argument slot
    // (a gain travels in thousandths); the log continues with the setter's own set_taps / set_gain / set_sensitivity / set_k events behind a marker line
    else if (k == "demod_nbfm.set_filter_width") { auto p = make_gr_demod_nbfm(sps, samp_rate, carrier_freq, filter_width); s.lines.push_back("== set_filter_width"); p->set_filter_width(fm); }
    else if (k == "demod_am.set_filter_width") { auto p = make_gr_demod_am(sps, samp_rate, carrier_freq, filter_width); s.lines.push_back("== set_filter_width"); p->set_filter_width(fm); }
    else if (k == "demod_wbfm.set_filter_width") { auto p = make_gr_demod_wbfm(sps, samp_rate, carrier_freq, filter_width); s.lines.push_back("== set_filter_width"); p->set_filter_width(fm); }
    else if (k == "demod_usb.set_filter_width") { auto p = make_gr_demod_ssb(sps, samp_rate, carrier_freq, filter_width, 0); s.lines.push_back("== set_filter_width"); p->set_filter_width(fm); }
    else if (k == "demod_lsb.set_filter_width") { auto p = make_gr_demod_ssb(sps, samp_rate, carrier_freq, filter_width, 1); s.lines.push_back("== set_filter_width"); p->set_filter_width(fm); }
    else if (k == "demod_usb.set_gain") { auto p = make_gr_demod_ssb(sps, samp_rate, carrier_freq, filter_width, 0); s.lines.push_back("== set_gain"); p->set_gain((float)fm / 1000.0f); }
    else if (k == "mod_nbfm.set_filter_width") { auto p = make_gr_mod_nbfm(sps, samp_rate, carrier_freq, filter_width); s.lines.push_back("== set_filter_width"); p->set_filter_width(fm); }
    else if (k == "mod_am.set_filter_width") { auto p = make_gr_mod_am(sps, samp_rate, carrier_freq, filter_width); s.lines.push_back("== set_filter_width"); p->set_filter_width(fm); }
    else if (k == "mod_usb.set_filter_width") { auto p = make_gr_mod_ssb(sps, samp_rate, carrier_freq, filter_width, 0); s.lines.push_back("== set_filter_width"); p->set_filter_width(fm); }
    else if (k == "mod_lsb.set_filter_width") { auto p = make_gr_mod_ssb(sps, samp_rate, carrier_freq, filter_width, 1); s.lines.push_back("== set_filter_width"); p->set_filter_width(fm); }
    else ok = false;
    if (!ok) return nullptr;
    g_text.clear();
    for (const std::string& l : s.lines) { g_text += l; g_text += '\n'; }
    return g_text.c_str();
}

}  // extern "C"
