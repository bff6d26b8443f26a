// ref_shim_gr.cpp -- TEST INFRASTRUCTURE.  extern "C" entry points into the reference's OWN GNU Radio custom blocks, compiled
// unmodified from where they lie (make -C oracle ref) against oracle/gr_stub (a stand-in for the few GNU Radio runtime classes they
// derive from).  Their work() functions hold the algorithms the oracle restates:
//   gr_dmr_dmo_sink::general_work   /root/reference/src/gr/gr_dmr_dmo_sink.cpp   (DMO correlator slicer)  -> orc_dmr.c
//   gr_deframer_bb::work            /root/reference/src/gr/gr_deframer_bb.cpp    (8(f) rank 1)                          -> orc_deframer
//   gr_4fsk_discriminator::work     /root/reference/src/gr/gr_4fsk_discriminator.cpp                                    -> orc_demod_4fsk
//   rssi_tag_block::work            /root/reference/src/gr/rssi_tag_block.cpp                                           -> orc_rssi_tag
//   dsss_decoder_cc_impl::general_work  /root/reference/src/gr/dsss_decoder_cc_impl.cc (its FIR kernel / RRC design come from gr_stub)  -> orc_dsss_decoder
//   cessb::clipper_cc / stretcher_cc    /root/reference/src/gr/cessb/*.cc (VOLK kernels from gr_stub/volk, generic forms)                 -> orc_cessb_*
//   gr_zero_idle_bursts::work       /root/reference/src/gr/gr_zero_idle_bursts.cpp (MMDVM TX)                           -> orc_zero_idle_bursts
#include <cstdint>
#include <cstring>

#include "src/gr/gr_dmr_dmo_sink.h"
#include "src/gr/gr_deframer_bb.h"
#include "src/gr/gr_4fsk_discriminator.h"
#include "src/gr/rssi_tag_block.h"
#include "src/gr/gr_zero_idle_bursts.h"
#include "src/gr/dsss_decoder_cc_impl.h"
#include "src/gr/dsss_encoder_bb_impl.h"
#include "src/gr/cessb/clipper_cc.h"
#include "src/gr/cessb/stretcher_cc.h"

extern "C" {

// the whole float stream through one sink in calls of `chunk` items; records of 40 bytes {frame type, FN, colour code, 0, 33 frame bytes, 0 0 0}
size_t ref_dmo_sink(const float* in, size_t n, size_t chunk, uint8_t* records, size_t cap)
{
    gr_dmr_dmo_sink_sptr s = make_gr_dmr_dmo_sink();
    size_t nrec = 0;
    for (size_t pos = 0; pos < n; pos += chunk) {
        const size_t m = n - pos < chunk ? n - pos : chunk;
        gr_vector_int ninput(1, (int)m);
        gr_vector_const_void_star ins(1, in + pos);
        gr_vector_void_star outs;
        s->general_work((int)m, ninput, ins, outs);
        for (DMRFrame& f : s->get_data()) {
            if (nrec >= cap) return nrec;
            uint8_t* r = records + 40 * nrec++;
            std::memset(r, 0, 40);
            r[0] = f.getFrameType(); r[1] = f.getFN(); r[2] = f.getColorCode();
            f.getData(r + 4);
        }
    }
    return nrec;
}

size_t ref_deframer(int type, const uint8_t* bits, size_t n, size_t chunk, uint8_t* out, size_t cap)
{
    gr_deframer_bb_sptr d = make_gr_deframer_bb(type);
    size_t no = 0;
    for (size_t pos = 0; pos < n; pos += chunk) {
        const size_t m = n - pos < chunk ? n - pos : chunk;
        gr_vector_const_void_star ins(1, bits + pos);
        gr_vector_void_star outs;
        d->work((int)m, ins, outs);
        std::vector<unsigned char>* v = d->get_data();
        if (v) {
            for (unsigned char c : *v) if (no < cap) out[no++] = c;
            delete v;
        }
    }
    return no;
}

void ref_4fsk_discriminator(const float* a, const float* b, const float* c, const float* d, size_t n, float* out /* 2 n */)
{
    gr_4fsk_discriminator_sptr k = make_gr_4fsk_discriminator();
    gr_vector_const_void_star ins{a, b, c, d};
    gr_vector_void_star outs{out};
    k->work((int)n, ins, outs);
}

// rssi_tag_block: returns the tags as (offset, dB) pairs; out = the pass-through copy
size_t ref_rssi_tag(const float* in /* 2 n */, size_t n, size_t chunk, float* out, uint64_t* offsets, float* db, size_t cap)
{
    rssi_tag_block_sptr t = make_rssi_tag_block();
    size_t nt = 0;
    for (size_t pos = 0; pos < n; pos += chunk) {
        const size_t m = n - pos < chunk ? n - pos : chunk;
        gr_vector_const_void_star ins(1, in + 2 * pos);
        gr_vector_void_star outs(1, out + 2 * pos);
        t->stub_written = pos;
        t->work((int)m, ins, outs);
        for (const gr::tag_t& g : t->stub_tags) if (nt < cap) { offsets[nt] = g.offset; db[nt] = pmt::to_float(g.value); ++nt; }
        t->stub_tags.clear();
    }
    return nt;
}

// gr_zero_idle_bursts (delay 0): "zero_samples" tags (offset, count) on the input stream
void ref_zero_idle_bursts(const float* in /* 2 n */, size_t n, size_t chunk, const uint64_t* offsets, const uint64_t* counts, size_t ntags, float* out)
{
    gr_zero_idle_bursts_sptr z = make_gr_zero_idle_bursts(0);
    for (size_t i = 0; i < ntags; ++i) z->stub_in_tags.push_back(gr::tag_t{offsets[i], pmt::string_to_symbol("zero_samples"), pmt::from_uint64(counts[i])});
    for (size_t pos = 0; pos < n; pos += chunk) {
        const size_t m = n - pos < chunk ? n - pos : chunk;
        gr_vector_const_void_star ins(1, in + 2 * pos);
        gr_vector_void_star outs(1, out + 2 * pos);
        z->stub_written = pos; z->stub_read = pos;
        z->work((int)m, ins, outs);
    }
}

// gr_zero_idle_bursts(delay > 0) in ONE work() call over the whole stream: the block asked for a history of 2 * SAMPLES_PER_SLOT items, so the
// scheduler hands it a window that starts 2 * 720 - 1 items (zeros at the start of a stream) in front of the first new item
void ref_zero_idle_bursts_delay(const float* in /* 2 n */, size_t n, unsigned delay, const uint64_t* offsets, const uint64_t* counts, size_t ntags, float* out)
{
    gr_zero_idle_bursts_sptr z = make_gr_zero_idle_bursts(delay);
    for (size_t i = 0; i < ntags; ++i) z->stub_in_tags.push_back(gr::tag_t{offsets[i], pmt::string_to_symbol("zero_samples"), pmt::from_uint64(counts[i])});
    const size_t H1 = delay > 0 ? 2 * 720 - 1 : 0;
    std::vector<gr_complex> buf(H1 + n, gr_complex(0, 0));
    std::memcpy(buf.data() + H1, in, n * sizeof(gr_complex));
    gr_vector_const_void_star ins(1, buf.data());
    gr_vector_void_star outs(1, out);
    z->stub_written = 0; z->stub_read = 0;
    z->work((int)n, ins, outs);
}

// gr::dsss::dsss_decoder_cc(Barker 13, sps): in = the whole stream; the block sees it with its history (set_history(13 sps): 13 sps - 1
// zeros in front of the first item) and is asked for every output whose input exists.  Returns the outputs; taps optional.
size_t ref_dsss_decoder(const float* in /* 2 n */, size_t n, int sps, size_t per_call, float* out /* 2 per output */, float* taps_out /* nt or NULL */)
{
    static const int barker_13[] = {1, 1, 1, 1, 1, 0, 0, 1, 1, 0, 1, 0, 1};
    std::vector<int> code(barker_13, barker_13 + 13);
    gr::dsss::dsss_decoder_cc::sptr d = gr::dsss::dsss_decoder_cc::make(code, (float)sps);
    const std::vector<gr_complex> taps = d->taps();
    if (taps_out) for (size_t i = 0; i < taps.size(); ++i) taps_out[i] = taps[i].real();
    const size_t L = 13 * (size_t)sps, nt = taps.size();
    // one more code period of zeros in front: output 0 reads in - L + j, i.e. BEFORE its history (GNU Radio's zero-filled circular
    // buffer hands out zeros there at the start of a stream)
    std::vector<gr_complex> buf(L + L - 1 + n + nt + L, gr_complex(0, 0));
    std::memcpy(buf.data() + L + (L - 1), in, n * sizeof(gr_complex));
    // output I reads in + (i - 1) L + j + [0, nt) with in = item (consumed - (L - 1)): it exists once item (I - 1) L + nt - 1 does
    const size_t total = n >= nt - L + L * 0 && n + L >= nt ? (n - (nt - L)) / L + 1 : 0;
    size_t done = 0, consumed = 0;
    while (done < total) {
        const size_t m = total - done < per_call ? total - done : per_call;
        gr_vector_int ninput(1, (int)(m * L));
        gr_vector_const_void_star ins(1, buf.data() + L + consumed);         // = item consumed - (L - 1)
        gr_vector_void_star outs(1, out + 2 * done);
        d->stub_consumed = 0;
        if (d->general_work((int)m, ninput, ins, outs) != (int)m) return 0;
        consumed += (size_t)d->stub_consumed;
        done += m;
    }
    return total;
}

// cessb::clipper_cc(clip): a sync block working in chunks of 1024 (n a multiple of 1024)
void ref_cessb_clipper(const float* in /* 2 n */, size_t n, float clip, float* out)
{
    gr::cessb::clipper_cc::sptr c = gr::cessb::clipper_cc::make(clip);
    gr_vector_const_void_star ins(1, in);
    gr_vector_void_star outs(1, out);
    c->work((int)n, ins, outs);
}
// cessb::stretcher_cc: general_work over whole chunks of 1024; the block reads two items beyond the chunk it writes (forecast), so
// n input items give 1024 floor((n - 2) / 1024) outputs.  calls of `chunks` chunks each.
size_t ref_cessb_stretcher(const float* in /* 2 n */, size_t n, size_t chunks, float* out)
{
    gr::cessb::stretcher_cc::sptr s = gr::cessb::stretcher_cc::make();
    const size_t total = n >= 2 ? 1024 * ((n - 2) / 1024) : 0;
    size_t done = 0;
    while (done < total) {
        const size_t m = total - done < 1024 * chunks ? total - done : 1024 * chunks;
        gr_vector_int ninput(1, (int)(n - done));
        gr_vector_const_void_star ins(1, in + 2 * done);
        gr_vector_void_star outs(1, out + 2 * done);
        if (s->general_work((int)m, ninput, ins, outs) != (int)m) return 0;
        done += m;
    }
    return total;
}

// gr::dsss::dsss_encoder_bb(Barker 13): packed bytes in, 13 chips per bit out
size_t ref_dsss_encoder(const uint8_t* in, size_t nbytes, uint8_t* out /* 104 per byte */)
{
    static const int barker_13[] = {1, 1, 1, 1, 1, 0, 0, 1, 1, 0, 1, 0, 1};
    std::vector<int> code(barker_13, barker_13 + 13);
    gr::dsss::dsss_encoder_bb::sptr e = gr::dsss::dsss_encoder_bb::make(code);
    gr_vector_int ninput(1, (int)nbytes);
    gr_vector_const_void_star ins(1, in);
    gr_vector_void_star outs(1, out);
    return (size_t)e->general_work((int)(nbytes * 104), ninput, ins, outs);
}

}
