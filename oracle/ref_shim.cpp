// ref_shim.cpp -- TEST INFRASTRUCTURE.  extern "C" entry points into the REAL reference sources of the frame FEC (compiled from
// where they lie under /root/reference by `make -C oracle ref`, output oracle/_ref/libqrl_ref.so; nothing of the reference is
// copied into this repository):
//   CBPTC19696::decode / encode           /root/reference/src/MMDVM/BPTC19696.cpp (+ Hamming.cpp, Utils.cpp, Log.cpp)
//   M17FrameDecoder::decodeFrame          /root/reference/src/M17/M17/M17FrameDecoder.cpp (+ M17Viterbi.hpp, M17Golay.cpp, ...)
//   M17FrameEncoder::encodeLsf / encodeStreamFrame   .../M17FrameEncoder.cpp (generates the test frames)
// Used by the tests to PIN the oracle's restatement (oracle/orc_framefec.c) and to regenerate tests/golden/ref/framefec.npz.
#include <cstdint>
#include <cstring>

#include "BPTC19696.h"
#include "src/gr/emphasis.h"
#include <M17/M17FrameDecoder.hpp>
#include <M17/M17FrameEncoder.hpp>
#include <M17/M17Golay.hpp>
#include <M17/M17Decorrelator.hpp>

extern "C" {

void ref_bptc19696_decode(const unsigned char* in33, unsigned char* out12)
{
    CBPTC19696 b;
    b.decode(in33, out12);
}
void ref_bptc19696_encode(const unsigned char* in12, unsigned char* inout33)
{
    CBPTC19696 b;
    b.encode(in12, inout33);
}

// one frame through a fresh decoder: type (M17FrameType value), the 30 LSF bytes and the 18 stream-frame bytes it holds afterwards
int ref_m17_decode_frame(const uint8_t frame[48], uint8_t lsf30[30], uint8_t stream18[18])
{
    M17::M17FrameDecoder d;
    d.reset();
    M17::frame_t f;
    std::memcpy(f.data(), frame, 48);
    const M17::M17FrameType t = d.decodeFrame(f);
    M17::M17LinkSetupFrame l = d.getLsf();
    M17::M17StreamFrame sf = d.getStreamFrame();
    std::memcpy(lsf30, l.getData(), 30);
    std::memcpy(stream18, sf.getData(), 18);
    return static_cast<int>(t);
}
// a sequence of frames through ONE decoder (LSF reassembly from the LICH segments): the LSF after the last frame
int ref_m17_decode_sequence(const uint8_t* frames, int n, uint8_t lsf30[30], uint8_t stream18[18])
{
    M17::M17FrameDecoder d;
    d.reset();
    int t = 0;
    for (int i = 0; i < n; ++i) {
        M17::frame_t f;
        std::memcpy(f.data(), frames + 48 * i, 48);
        t = static_cast<int>(d.decodeFrame(f));
    }
    M17::M17LinkSetupFrame l = d.getLsf();
    M17::M17StreamFrame sf = d.getStreamFrame();
    std::memcpy(lsf30, l.getData(), 30);
    std::memcpy(stream18, sf.getData(), 18);
    return t;
}
// LSF with the given 28 bytes (dst, src, type, meta; the CRC is computed) -> LSF frame, then nstream stream frames of payloads
void ref_m17_encode(const uint8_t lsf28[28], const uint8_t* payloads /* [nstream][16] */, int nstream, uint8_t* frames /* [1 + nstream][48] */)
{
    M17::M17LinkSetupFrame lsf;
    std::memcpy(const_cast<uint8_t*>(lsf.getData()), lsf28, 28);
    lsf.updateCrc();
    M17::M17FrameEncoder e;
    e.reset();
    M17::frame_t f;
    e.encodeLsf(lsf, f);
    std::memcpy(frames, f.data(), 48);
    for (int i = 0; i < nstream; ++i) {
        M17::payload_t p;
        std::memcpy(p.data(), payloads + 16 * i, 16);
        e.encodeStreamFrame(p, f, i == nstream - 1);
        std::memcpy(frames + 48 * (i + 1), f.data(), 48);
    }
}
// gr::calculate_deemph_taps (reference src/gr/emphasis.cpp:16-43): a = {1, -p1}, b = {b0, b0}
void ref_deemph_taps(int sample_rate, double tau, double a[2], double b[2])
{
    std::vector<double> at, bt;
    gr::calculate_deemph_taps(sample_rate, tau, at, bt);
    a[0] = at[0]; a[1] = at[1]; b[0] = bt[0]; b[1] = bt[1];
}
void ref_preemph_taps(int sample_rate, double tau, double a[2], double b[2])
{
    std::vector<double> at, bt;
    gr::calculate_preemph_taps(sample_rate, tau, at, bt);
    a[0] = at[0]; a[1] = at[1]; b[0] = bt[0]; b[1] = bt[1];
}
uint32_t ref_golay24_encode(uint16_t data) { return M17::golay24_encode(data); }
uint16_t ref_golay24_decode(uint32_t cw) { return M17::golay24_decode(cw); }
void ref_m17_decorrelator_sequence(uint8_t out46[46])
{
    std::array<uint8_t, 46> z;
    z.fill(0);
    M17::decorrelate(z);
    std::memcpy(out46, z.data(), 46);
}

}
