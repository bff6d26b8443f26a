// kernels_dmo.hip — the DMR DMO correlator slicer on the device.
//
//  k_dmo_sink : gr_dmr_dmo_sink::general_work / processSample / correlateSync / samplesToBits
//               [reference src/gr/gr_dmr_dmo_sink.cpp:63-79, 81-204, 206-322, 324-357; constants src/DMR/constants.h:8-36,71-88;
//                slot-type decode src/MMDVM/DMRSlotType2.cpp:240-264 (Golay (20,8) through the (19,8) syndrome table)]
//
// The block is sample-serial per stream (a shift register of sign bits per sampling phase, a Hamming test against the two MS
// sync words, a 24-point correlation when the test passes, then 132 symbols sliced with the averaged centre / threshold), so
// parallelism comes from the batch: ONE LANE PER STREAM.  The reference's 1440-sample ring m_buffer is not copied: the kernel
// reads the engine ring that holds port 3 of gr_demod_dmr (RRC-filtered discriminator output, absolute sample index), slot p of
// m_buffer at sample n being absolute index n - ((n - p) mod 1440) -- including the reference's read of slot m_endPtr + 1, which
// is the sample of one lap earlier (tests/test_dmo_sink.py).  Frames leave as 40-byte records {frame type, FN, colour code, 0,
// 33 frame bytes, 3 pad}; the arithmetic (float compares, one float multiply-add chain of 24 terms) is that of oracle/orc_dmr.c.
#include "devmath.hpp"
#include "engine.hpp"

namespace qrl {

namespace {
constexpr int DMO_BUF = 1440, SYM = 5, FRAME_BYTES = 33, FRAME_SYMBOLS = 132, FRAME_SAMPLES = 660, SYNC_SYMBOLS = 24, SYNC_SAMPLES = 120;
constexpr int SLOT_TYPE_SAMPLES = 50, INFO_SAMPLES = 490, NOENDPTR = 9999;
__constant__ int8_t c_data_values[24] = {-3, +3, +3, +3, -3, +3, +3, -3, -3, -3, +3, -3, +3, -3, -3, -3, -3, +3, +3, -3, +3, +3, +3, -3};
__constant__ int8_t c_voice_values[24] = {+3, -3, -3, -3, +3, -3, -3, +3, +3, +3, -3, +3, -3, +3, +3, +3, +3, -3, -3, +3, -3, -3, -3, +3};
__constant__ uint8_t c_data_bytes[7] = {0x0D, 0x5D, 0x7F, 0x77, 0xFD, 0x75, 0x70};
__constant__ uint8_t c_voice_bytes[7] = {0x07, 0xF7, 0xD5, 0xDD, 0x57, 0xDF, 0xD0};
__constant__ uint8_t c_sync_mask[7] = {0x0F, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xF0};

struct Ctx {
    const float* row; uint32_t mask; int64_t n;   // stream's ring row, current absolute sample index
    __device__ float buf(unsigned p) const {      // m_buffer[p] as of sample n (dataPtr = n mod 1440)
        const unsigned dp = (unsigned)(n % DMO_BUF);
        const unsigned back = (dp + DMO_BUF - p) % DMO_BUF;
        const int64_t i = n - back;
        return i >= 0 ? row[(uint32_t)i & mask] : 0.0f;
    }
};
__device__ uint32_t golay_syndrome(uint32_t pattern)
{
    uint32_t aux = 0x40000u;
    if (pattern >= 0x800u) {
        while (pattern & 0xFFFFF800u) {
            while (!(aux & pattern)) aux >>= 1;
            pattern ^= (aux / 0x800u) * 0xC75u;
        }
    }
    return pattern;
}
__device__ void samples_to_bits(const Ctx& c, unsigned start, unsigned count, uint8_t* buffer, unsigned offset, float centre, float threshold)
{
    for (unsigned i = 0; i < count; i++) {
        const float sample = c.buf(start) - centre;
        int b0, b1;
        if (sample < -threshold) { b0 = 1; b1 = 1; }
        else if (sample < 0.0f) { b0 = 1; b1 = 0; }
        else if (sample < threshold) { b0 = 0; b1 = 0; }
        else { b0 = 0; b1 = 1; }
        const uint8_t m0 = (uint8_t)(0x80u >> (offset & 7));
        buffer[offset >> 3] = b0 ? (uint8_t)(buffer[offset >> 3] | m0) : (uint8_t)(buffer[offset >> 3] & ~m0);
        offset++;
        const uint8_t m1 = (uint8_t)(0x80u >> (offset & 7));
        buffer[offset >> 3] = b1 ? (uint8_t)(buffer[offset >> 3] | m1) : (uint8_t)(buffer[offset >> 3] & ~m1);
        offset++;
        start += SYM;
        if (start >= DMO_BUF) start -= DMO_BUF;
    }
}
__device__ void dmo_reset(DmoState& s)
{
    s.syncPtr = 0; s.maxCorr = 0; s.syncCount = 0; s.state = 0; s.startPtr = 0; s.endPtr = NOENDPTR; s.colorCode = 0; s.n = 0;
}
__device__ void correlate_sync(DmoState& s, const Ctx& c, unsigned dataPtr, unsigned bitPtr, bool first)
{
    const uint32_t sh = s.bitBuffer[bitPtr] & 0x00FFFFFFu;
    const bool data = __popc(sh ^ 0x0076286Eu) <= 2, voice = __popc(sh ^ 0x0089D791u) <= 2;
    if (!(data || voice)) return;
    unsigned ptr = dataPtr + DMO_BUF - SYNC_SAMPLES + SYM;
    if (ptr >= DMO_BUF) ptr -= DMO_BUF;
    float corr = 0.0f, mn = 100.0f, mx = -100.0f;
    unsigned p = ptr;
    for (int i = 0; i < SYNC_SYMBOLS; i++) {
        const float val = c.buf(p);
        if (val > mx) mx = val;
        if (val < mn) mn = val;
        corr += (float)(data ? c_data_values[i] : c_voice_values[i]) * val;
        p += SYM;
        if (p >= DMO_BUF) p -= DMO_BUF;
    }
    if (!(corr > s.maxCorr)) return;
    const float centre = (mx + mn) / 2.0f;
    const float threshold = (mx - centre) / 2.0f;
    uint8_t sync[7] = {0, 0, 0, 0, 0, 0, 0};
    samples_to_bits(c, ptr, SYNC_SYMBOLS, sync, 4, centre, threshold);
    unsigned errs = 0;
    for (int i = 0; i < 7; i++) errs += __popc((uint32_t)((sync[i] & c_sync_mask[i]) ^ (data ? c_data_bytes[i] : c_voice_bytes[i])));
    if (errs > 3) return;
    if (first) {
        for (int i = 0; i < 4; i++) { s.threshold[i] = threshold; s.centre[i] = centre; }
        s.averagePtr = 0;
    } else {
        s.threshold[s.averagePtr] = threshold;
        s.centre[s.averagePtr] = centre;
        if (++s.averagePtr >= 4) s.averagePtr = 0;
    }
    s.maxCorr = corr;
    s.control = data ? 0x40 : 0x20;
    s.syncPtr = (uint16_t)dataPtr;
    unsigned sp = dataPtr + DMO_BUF - SLOT_TYPE_SAMPLES / 2 - INFO_SAMPLES / 2 - SYNC_SAMPLES;
    if (sp >= DMO_BUF) sp -= DMO_BUF;
    s.startPtr = (uint16_t)sp;
    unsigned ep = dataPtr + SLOT_TYPE_SAMPLES / 2 + INFO_SAMPLES / 2 - 1;
    if (ep >= DMO_BUF) ep -= DMO_BUF;
    s.endPtr = (uint16_t)ep;
}
}  // namespace

__global__ __launch_bounds__(64) void k_dmo_sink(const DmoParams P, int batch)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= batch) return;
    DmoState s = P.st[b];
    Ctx c{P.in.p + (size_t)b * (P.in.mask + 1u), P.in.mask, 0};
    uint8_t* out = P.out + (size_t)b * P.cap * 40;
    uint32_t nout = 0;
    enum { TYPE_DATA = 0, TYPE_VOICE = 1, TYPE_VOICE_SYNC = 2, RECV_NONE = 0, RECV_DATA = 1, RECV_VOICE = 3 };
    auto write_frame = [&](const uint8_t* frame, uint8_t type) {
        if (nout < P.cap) {
            uint8_t* r = out + 40 * (size_t)nout;
            r[0] = type; r[1] = s.n; r[2] = s.colorCode; r[3] = 0;
            for (int i = 0; i < FRAME_BYTES; ++i) r[4 + i] = frame[i];
            r[37] = r[38] = r[39] = 0;
        }
        ++nout;
    };
    for (uint32_t k = 0; k < P.count; ++k) {
        c.n = (int64_t)(P.q0 + k);
        const unsigned dataPtr = (unsigned)(c.n % DMO_BUF), bitPtr = (unsigned)(c.n % SYM);
        const float sample = c.row[(uint32_t)c.n & c.mask];
        s.bitBuffer[bitPtr] <<= 1;
        if (sample > 0.0f) s.bitBuffer[bitPtr] |= 1u;
        if (s.state == RECV_NONE) correlate_sync(s, c, dataPtr, bitPtr, true);
        else {
            unsigned mn = s.syncPtr + DMO_BUF - 1, mx = s.syncPtr + 1;
            if (mn >= DMO_BUF) mn -= DMO_BUF;
            if (mx >= DMO_BUF) mx -= DMO_BUF;
            if (mn < mx) { if (dataPtr >= mn && dataPtr <= mx) correlate_sync(s, c, dataPtr, bitPtr, false); }
            else { if (dataPtr >= mn || dataPtr <= mx) correlate_sync(s, c, dataPtr, bitPtr, false); }
        }
        if (dataPtr == s.endPtr) {
            const float centre = (s.centre[0] + s.centre[1] + s.centre[2] + s.centre[3]) / 4.0f;
            const float threshold = (s.threshold[0] + s.threshold[1] + s.threshold[2] + s.threshold[3]) / 4.0f;
            uint8_t frame[FRAME_BYTES];
            for (int i = 0; i < FRAME_BYTES; ++i) frame[i] = 0;
            unsigned ptr = s.endPtr + DMO_BUF - FRAME_SAMPLES + SYM + 1;
            if (ptr >= DMO_BUF) ptr -= DMO_BUF;
            samples_to_bits(c, ptr, FRAME_SYMBOLS, frame, 0, centre, threshold);
            if (s.control == 0x40) {
                uint8_t st[3];
                st[0] = (uint8_t)(((frame[12] << 2) & 0xFC) | ((frame[13] >> 6) & 0x03));
                st[1] = (uint8_t)(((frame[13] << 2) & 0xC0) | ((frame[19] << 2) & 0x3C) | ((frame[20] >> 6) & 0x03));
                st[2] = (uint8_t)((frame[20] << 2) & 0xF0);
                uint32_t code = ((uint32_t)st[0] << 11) + ((uint32_t)st[1] << 3) + ((uint32_t)st[2] >> 5);
                const uint32_t e = P.golay[golay_syndrome(code)];
                if (e) code ^= e;
                const uint8_t cw = (uint8_t)(code >> 11), dataType = cw & 0x0F;
                s.colorCode = (cw >> 4) & 0x0F;
                s.syncCount = 0; s.n = 0;
                switch (dataType) {
                case 0x06: s.state = RECV_DATA; write_frame(frame, TYPE_DATA); break;
                case 0x07: case 0x08: case 0x0A: if (s.state == RECV_DATA) write_frame(frame, TYPE_DATA); break;
                case 0x01: case 0x00: s.state = RECV_VOICE; write_frame(frame, TYPE_DATA); break;
                case 0x02: if (s.state == RECV_VOICE) { write_frame(frame, TYPE_DATA); dmo_reset(s); } break;
                default: write_frame(frame, TYPE_DATA); dmo_reset(s); break;
                }
            } else if (s.control == 0x20) {
                s.state = RECV_VOICE; s.syncCount = 0; s.n = 0;
                write_frame(frame, TYPE_VOICE_SYNC);
            } else {
                if (s.state != RECV_NONE) {
                    s.syncCount++;
                    if (s.syncCount >= 13) dmo_reset(s);
                }
                if (s.state == RECV_VOICE) {
                    if (s.n >= 5) s.n = 0; else ++s.n;
                    write_frame(frame, TYPE_VOICE);
                } else if (s.state == RECV_DATA) {
                    write_frame(frame, TYPE_DATA);
                }
            }
            s.maxCorr = 0;
            s.control = 0;
        }
    }
    P.st[b] = s;
    P.counts[b] = nout;   // bursts FOUND by the call; records beyond cap_frames are not written: counts[b] > cap tells the caller that bursts were dropped
}

void launch_dmo_sink(const DmoParams& p, int batch, hipStream_t s)
{
    if (!p.count) { (void)hipMemsetAsync(p.counts, 0, (size_t)batch * sizeof(uint32_t), s); return; }
    hipLaunchKernelGGL(k_dmo_sink, dim3((batch + 63) / 64), dim3(64), 0, s, p, batch);
}

// DECODING_TABLE_1987 of DMRSlotType2.cpp:46-205, generated: error patterns of weight 1..5 over the 19 code bits in lexicographic
// order of their bit positions, first pattern per syndrome wins (the rule reproduces all 2048 literals: tests/test_dmo_sink.py
// checks the oracle's identical generator against the reference source, tests/test_gpu_dmo.py this table against the oracle's)
static uint32_t host_syndrome(uint32_t pattern)
{
    uint32_t aux = 0x40000u;
    if (pattern >= 0x800u) {
        while (pattern & 0xFFFFF800u) {
            while (!(aux & pattern)) aux >>= 1;
            pattern ^= (aux / 0x800u) * 0xC75u;
        }
    }
    return pattern;
}
static void golay_enum(std::vector<uint32_t>& t, int weight, int from, uint32_t e)
{
    if (weight == 0) { const uint32_t s = host_syndrome(e); if (s && !t[s]) t[s] = e; return; }
    for (int b = from; b < 19; ++b) golay_enum(t, weight - 1, b + 1, e | (1u << b));
}
std::vector<uint32_t> golay1987_table()
{
    std::vector<uint32_t> t(2048, 0u);
    for (int w = 1; w <= 5; ++w) golay_enum(t, w, 0, 0);
    return t;
}

}  // namespace qrl
