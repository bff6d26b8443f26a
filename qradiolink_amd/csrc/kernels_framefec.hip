// kernels_framefec.hip — frame FEC of the DMR and M17 protocol stacks over batches of frames:
//   k_bptc_decode / k_bptc_encode   CBPTC19696::decode / encode   reference src/MMDVM/BPTC19696.cpp:47-87, Hamming (15,11,3) and
//                                   (13,9,3) src/MMDVM/Hamming.cpp:79-180
//   k_m17_decode                    M17FrameDecoder::decodeFrame   reference src/M17/M17/M17FrameDecoder.cpp:44-215 (decorrelator,
//                                   quadratic de-interleaver, sync-word classification, punctured K = 5 Viterbi M17Viterbi.hpp:96-221,
//                                   Golay(24,12) of the LICH M17Golay.cpp:24-95)
// Integer / bit work, one thread per frame.  BPTC: the 13 x 15 code matrix lives in 13 registers (one row each), the column code is
// decoded bit-sliced for all 15 columns at once (four syndrome planes, thirteen flip masks), the row code by syndrome -> column match.
// M17: 16 path metrics in registers, the decision words of the (at most 244) trellis steps in LDS, [step][thread].
// Results = oracle/orc_framefec.c, which is pinned against the reference's own sources (oracle/_ref).
#include "engine.hpp"

namespace qrl {

// parity equations over the data bits (ETSI TS 102 361-1 B.3): bit j of mask p set = data bit j enters parity p
__constant__ uint16_t c_h15[4] = {0x01AF, 0x035E, 0x06BC, 0x04D7};
__constant__ uint16_t c_h13[4] = {0x006B, 0x00D7, 0x01AF, 0x0135};

__device__ __forceinline__ unsigned par32(unsigned v) { return (unsigned)__popc(v) & 1u; }

// syndrome of a (k + 4)-bit word, and the word with the single error that syndrome names flipped (none if it names no column)
__device__ __forceinline__ unsigned hamming_fix(unsigned w, const uint16_t* mask, int k, bool& fixed)
{
    unsigned syn = 0;
#pragma unroll
    for (int p = 0; p < 4; ++p) syn |= (par32(w & mask[p]) ^ ((w >> (k + p)) & 1u)) << p;
    fixed = false;
    if (!syn) return w;
    for (int j = 0; j < k + 4; ++j) {
        unsigned col;
        if (j < k) col = ((mask[0] >> j) & 1u) | (((mask[1] >> j) & 1u) << 1) | (((mask[2] >> j) & 1u) << 2) | (((mask[3] >> j) & 1u) << 3);
        else col = 1u << (j - k);
        if (col == syn) { fixed = true; return w ^ (1u << j); }
    }
    return w;
}

// transmitted bit i (0 .. 195) of a 33-byte burst: bytes 0..11, the top two bits of byte 12, the low two bits of byte 20, bytes 21..32
__device__ __forceinline__ unsigned burst_bit(const uint8_t* in, int i)
{
    if (i < 98) return (in[i >> 3] >> (7 - (i & 7))) & 1u;
    if (i < 100) return (in[20] >> (99 - i)) & 1u;
    const int k = i - 100;
    return (in[21 + (k >> 3)] >> (7 - (k & 7))) & 1u;
}

__global__ __launch_bounds__(64) void k_bptc_decode(const uint8_t* __restrict__ bursts, size_t n, uint8_t* __restrict__ payloads)
{
    const size_t f = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (f >= n) return;
    uint8_t in[33];
    for (int i = 0; i < 33; ++i) in[i] = bursts[f * 33 + i];
    // de-interleave: matrix bit a (1 .. 195; bit 0 = R(3), unused) = transmitted bit (181 a) mod 196; row r holds bits 1 + 15 r ..
    unsigned row[13];
#pragma unroll
    for (int r = 0; r < 13; ++r) {
        unsigned w = 0;
        for (int j = 0; j < 15; ++j) w |= burst_bit(in, ((1 + 15 * r + j) * 181) % 196) << j;
        row[r] = w;
    }
    bool fixing;
    int count = 0;
    do {
        fixing = false;
        // columns, Hamming (13,9,3), all 15 at once: plane p = the p-th syndrome bit of every column
        unsigned s[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            unsigned v = row[9 + p];
#pragma unroll
            for (int a = 0; a < 9; ++a) if ((c_h13[p] >> a) & 1u) v ^= row[a];
            s[p] = v;
        }
#pragma unroll
        for (int a = 0; a < 13; ++a) {
            unsigned m = 0x7FFFu;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const unsigned bit = a < 9 ? (c_h13[p] >> a) & 1u : (unsigned)(a - 9 == p);
                m &= bit ? s[p] : ~s[p];
            }
            row[a] ^= m;
            fixing |= m != 0;
        }
        // the 9 rows that carry data, Hamming (15,11,3)
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            bool fx;
            row[r] = hamming_fix(row[r], c_h15, 11, fx);
            fixing |= fx;
        }
        ++count;
    } while (fixing && count < 5);
    // payload: row 0 bits 3..10, rows 1..8 bits 0..10, MSB first
    uint64_t lo = 0;   // 96 bits as 64 + 32
    unsigned hi = 0;
    int pos = 0;
    auto put = [&](unsigned bit) { if (pos < 64) lo |= (uint64_t)bit << (63 - pos); else hi |= bit << (95 - pos); ++pos; };
    for (int j = 3; j <= 10; ++j) put((row[0] >> j) & 1u);
    for (int r = 1; r <= 8; ++r) for (int j = 0; j <= 10; ++j) put((row[r] >> j) & 1u);
    for (int b = 0; b < 8; ++b) payloads[f * 12 + b] = (uint8_t)(lo >> (56 - 8 * b));
    for (int b = 0; b < 4; ++b) payloads[f * 12 + 8 + b] = (uint8_t)(hi >> (24 - 8 * b));
}

__global__ __launch_bounds__(64) void k_bptc_encode(const uint8_t* __restrict__ payloads, size_t n, uint8_t* __restrict__ bursts)
{
    const size_t f = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (f >= n) return;
    uint8_t in[12];
    for (int i = 0; i < 12; ++i) in[i] = payloads[f * 12 + i];
    auto bit = [&](int i) { return (unsigned)(in[i >> 3] >> (7 - (i & 7))) & 1u; };
    unsigned row[13];
    int pos = 0;
    {
        unsigned w = 0;
        for (int j = 3; j <= 10; ++j) w |= bit(pos++) << j;
        row[0] = w;
    }
    for (int r = 1; r <= 8; ++r) { unsigned w = 0; for (int j = 0; j <= 10; ++j) w |= bit(pos++) << j; row[r] = w; }
#pragma unroll
    for (int r = 0; r < 9; ++r)
#pragma unroll
        for (int p = 0; p < 4; ++p) row[r] |= par32(row[r] & c_h15[p]) << (11 + p);
#pragma unroll
    for (int p = 0; p < 4; ++p) {      // column parities, bit-sliced: parity row 9 + p = XOR of the data rows of its equation
        unsigned v = 0;
#pragma unroll
        for (int a = 0; a < 9; ++a) if ((c_h13[p] >> a) & 1u) v ^= row[a];
        row[9 + p] = v;
    }
    // interleave into the burst: transmitted bit (181 a) mod 196 = matrix bit a, matrix bit 0 = 0; the burst's other bits are kept
    uint8_t out[33];
    for (int i = 0; i < 33; ++i) out[i] = bursts[f * 33 + i];
    for (int i = 0; i < 12; ++i) { out[i] = 0; out[21 + i] = 0; }
    out[12] &= 0x3Fu; out[20] &= 0xFCu;
    for (int a = 1; a < 196; ++a) {
        const unsigned b = (row[(a - 1) / 15] >> ((a - 1) % 15)) & 1u;
        const int t = (a * 181) % 196;
        if (t < 98) out[t >> 3] |= (uint8_t)(b << (7 - (t & 7)));
        else if (t < 100) out[20] |= (uint8_t)(b << (99 - t));
        else out[21 + ((t - 100) >> 3)] |= (uint8_t)(b << (7 - ((t - 100) & 7)));
    }
    for (int i = 0; i < 33; ++i) bursts[f * 33 + i] = out[i];
}
void launch_bptc_decode(const uint8_t* bursts, size_t n, uint8_t* payloads, hipStream_t s)
{
    if (!n) return;
    hipLaunchKernelGGL(k_bptc_decode, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, bursts, n, payloads);
}
void launch_bptc_encode(const uint8_t* payloads, size_t n, uint8_t* bursts, hipStream_t s)
{
    if (!n) return;
    hipLaunchKernelGGL(k_bptc_encode, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, payloads, n, bursts);
}

// ------------------------------------------------------------------------------------------------------------------ M17
__constant__ uint8_t c_m17_seq[46] = {
    0xD6, 0xB5, 0xE2, 0x30, 0x82, 0xFF, 0x84, 0x62, 0xBA, 0x4E, 0x96, 0x90, 0xD8, 0x98, 0xDD, 0x5D, 0x0C, 0xC8, 0x52, 0x43, 0x91, 0x1D, 0xF8,
    0x6E, 0x68, 0x2F, 0x35, 0xDA, 0x14, 0xEA, 0xCD, 0x76, 0x19, 0x8D, 0xD5, 0x80, 0xD1, 0x33, 0x87, 0x13, 0x57, 0x18, 0x2D, 0x29, 0x78, 0xC3};
// Golay(24,12) with generator polynomial 0xC75, tables by rule at compile time: checksum of data bit i = (x^(i + 11) mod g) << 1 | the
// overall parity of the 23-bit word; dec = the inverse map (parity error pattern -> data error pattern) by Gaussian elimination
struct GolayTables { uint16_t enc[12], dec[12]; };
constexpr GolayTables make_golay()
{
    GolayTables t{};
    for (int i = 0; i < 12; ++i) {
        uint32_t v = (uint32_t)1 << (i + 11);
        for (int b = 22; b >= 11; --b) if (v & ((uint32_t)1 << b)) v ^= (uint32_t)0xC75 << (b - 11);
        uint32_t cw = ((uint32_t)1 << (i + 11)) | v, par = 0;
        for (int b = 0; b < 23; ++b) par ^= (cw >> b) & 1u;
        t.enc[i] = (uint16_t)((v << 1) | par);
    }
    uint32_t m[12] = {};
    for (int i = 0; i < 12; ++i) m[i] = ((uint32_t)t.enc[i] << 12) | ((uint32_t)1 << i);
    for (int c = 0; c < 12; ++c) {
        int p = c;
        while (!(m[p] & ((uint32_t)1 << (12 + c)))) ++p;
        const uint32_t sw = m[p]; m[p] = m[c]; m[c] = sw;
        for (int r = 0; r < 12; ++r) if (r != c && (m[r] & ((uint32_t)1 << (12 + c)))) m[r] ^= m[c];
    }
    for (int c = 0; c < 12; ++c) t.dec[c] = (uint16_t)(m[c] & 0xFFFu);
    return t;
}
__constant__ GolayTables c_gol = make_golay();
#define c_gol_enc c_gol.enc
#define c_gol_dec c_gol.dec

__device__ unsigned golay24_decode(unsigned cw)   // 0xFFFF: uncorrectable
{
    const unsigned data = (cw >> 12) & 0xFFFu, parity = cw & 0xFFFu;
    unsigned chk = 0;
    for (int i = 0; i < 12; ++i) if (data & (1u << i)) chk ^= c_gol_enc[i];
    const unsigned syn = parity ^ chk;
    if (__popc(syn) <= 3) return ((cw ^ syn) >> 12) & 0xFFFu;
    for (int i = 0; i < 12; ++i)
        if (__popc(syn ^ c_gol_enc[i]) <= 2) return (data ^ (1u << i)) & 0xFFFu;
    unsigned inv = 0;
    for (int i = 0; i < 12; ++i) if (syn & (1u << i)) inv ^= c_gol_dec[i];
    if (__popc(inv) <= 3) return (data ^ inv) & 0xFFFu;
    for (int i = 0; i < 12; ++i)
        if (__popc(inv ^ c_gol_dec[i]) <= 2) return (data ^ inv ^ c_gol_dec[i]) & 0xFFFu;
    return 0xFFFFu;
}

constexpr int M17_TPB = 64, M17_MAXSTEPS = 244;

// punctured hard-decision Viterbi (K = 5, G1 = 0x19, G2 = 0x17) of nbits received bits starting at bit `first` of di[]; the puncture
// matrix is "every fourth entry, starting with the third, is dropped" over 61 entries (LSF) or "the twelfth of twelve" (stream)
__device__ void m17_viterbi(const uint8_t* di, int first, int nbits, bool lsf, uint16_t* hist /* [step * M17_TPB] */, uint8_t* out, int nbytes)
{
    unsigned prev[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) prev[i] = 0;
    const int P = lsf ? 61 : 12;
    int steps = 0, pi = 0, bp = 0;
    while (bp < nbits) {
        int sym[2] = {1, 1};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const bool keep = lsf ? (pi & 3) != 2 : pi != 11;
            ++pi;
            if (keep) { const int b = first + bp; sym[k] = ((di[b >> 3] >> (7 - (b & 7))) & 1) ? 2 : 0; ++bp; }
            if (pi >= P) pi = 0;
        }
        unsigned cur[16], h = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c0 = i >= 4 ? 2 : 0, c1 = ((i + 1) & 2) ? 2 : 0;       // expected pair of predecessor i on input 0
            const unsigned metric = (unsigned)(abs(c0 - sym[0]) + abs(c1 - sym[1]));
            const unsigned m0 = prev[i] + metric, m1 = prev[i + 8] + (4 - metric);
            const unsigned m2 = prev[i] + (4 - metric), m3 = prev[i + 8] + metric;
            if (m0 >= m1) { h |= 1u << (2 * i); cur[2 * i] = m1; } else cur[2 * i] = m0;
            if (m2 >= m3) { h |= 1u << (2 * i + 1); cur[2 * i + 1] = m3; } else cur[2 * i + 1] = m2;
        }
        hist[steps * M17_TPB] = (uint16_t)h;
        ++steps;
#pragma unroll
        for (int i = 0; i < 16; ++i) prev[i] = cur[i] & 0xFFFFu;   // the reference's metrics are 16-bit
    }
    unsigned state = 0;
    int pos = steps;
    for (int b = nbytes * 8 - 1; b >= 0; --b) {
        --pos;
        const unsigned bit = (hist[pos * M17_TPB] >> (state >> 4)) & 1u;
        state = (state >> 1) | (bit << 7);
        if (bit) out[b >> 3] |= (uint8_t)(0x80u >> (b & 7));
    }
}

__global__ __launch_bounds__(M17_TPB) void k_m17_decode(const uint8_t* __restrict__ frames, size_t n, uint8_t* __restrict__ records)
{
    __shared__ uint16_t hist[M17_MAXSTEPS * M17_TPB];
    const size_t f = (size_t)blockIdx.x * M17_TPB + threadIdx.x;
    if (f >= n) return;
    uint8_t data[46], di[46], rec[40];
    const uint8_t s0 = frames[f * 48], s1 = frames[f * 48 + 1];
    for (int i = 0; i < 46; ++i) { data[i] = frames[f * 48 + 2 + i] ^ c_m17_seq[i]; di[i] = 0; }
    for (int i = 0; i < 40; ++i) rec[i] = 0;
    for (unsigned i = 0; i < 368; ++i) {
        const unsigned src = (45u * i + 92u * i * i) % 368u;
        if ((data[src >> 3] >> (7 - (src & 7))) & 1u) di[i >> 3] |= (uint8_t)(0x80u >> (i & 7));
    }
    // nearest sync word: preamble 0x7777, link setup 0x55F7, stream 0xFF5D; strictly smaller distance wins; > 4 bit errors: unknown
    int type = 0, best = __popc((unsigned)(s0 ^ 0x77)) + __popc((unsigned)(s1 ^ 0x77));
    int d = __popc((unsigned)(s0 ^ 0x55)) + __popc((unsigned)(s1 ^ 0xF7));
    if (d < best) { best = d; type = 1; }
    d = __popc((unsigned)(s0 ^ 0xFF)) + __popc((unsigned)(s1 ^ 0x5D));
    if (d < best) { best = d; type = 2; }
    if (best > 4) type = 4;
    rec[0] = (uint8_t)type;
    uint16_t* h = hist + threadIdx.x;
    if (type == 1) m17_viterbi(di, 0, 368, true, h, rec + 2, 30);
    if (type == 2) {
        unsigned dec[4];
        bool ok = true;
        for (int i = 0; i < 4 && ok; ++i) {
            dec[i] = golay24_decode(((unsigned)di[3 * i] << 16) | ((unsigned)di[3 * i + 1] << 8) | di[3 * i + 2]);
            ok = dec[i] != 0xFFFFu;
        }
        if (ok) {
            rec[32] = (uint8_t)(dec[0] >> 4); rec[33] = (uint8_t)(((dec[0] & 0xFu) << 4) | (dec[1] >> 8)); rec[34] = (uint8_t)(dec[1] & 0xFFu);
            rec[35] = (uint8_t)(dec[2] >> 4); rec[36] = (uint8_t)(((dec[2] & 0xFu) << 4) | (dec[3] >> 8)); rec[37] = (uint8_t)((dec[3] & 0xFFu) >> 5);
            rec[1] = 1;
        }
        m17_viterbi(di, 96, 272, false, h, rec + 2, 18);
    }
    for (int i = 0; i < 40; ++i) records[f * 40 + i] = rec[i];
}
// M17FrameEncoder::encodeLsf / encodeStreamFrame (reference src/M17/M17/M17FrameEncoder.cpp:52-118), stateless: a record as
// k_m17_decode writes it -> K = 5 convolutional code (+ four flush bits), puncturing, Golay(24,12) of the LICH blocks, quadratic
// interleaver, randomiser, sync word.  Thread per frame.
__global__ __launch_bounds__(64) void k_m17_encode(const uint8_t* __restrict__ records, size_t n, uint8_t* __restrict__ frames)
{
    const size_t f = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (f >= n) return;
    uint8_t rec[40], body[46], il[46];
    for (int i = 0; i < 40; ++i) rec[i] = records[f * 40 + i];
    for (int i = 0; i < 46; ++i) { body[i] = 0; il[i] = 0; }
    const bool lsf = rec[0] == 1;
    int ob = 0;
    if (!lsf) {
        const unsigned num = rec[37];
        const unsigned blocks[4] = {((unsigned)rec[32] << 4) | (rec[33] >> 4), (((unsigned)rec[33] & 0xFu) << 8) | rec[34],
                                    ((unsigned)rec[35] << 4) | (rec[36] >> 4), (((unsigned)rec[36] & 0xFu) << 8) | ((num << 5) & 0xFFu)};
        for (int i = 0; i < 4; ++i) {
            unsigned chk = 0;
            for (int k = 0; k < 12; ++k) if (blocks[i] & (1u << k)) chk ^= c_gol_enc[k];
            const unsigned cw = (blocks[i] << 12) | chk;
            body[3 * i] = (uint8_t)(cw >> 16); body[3 * i + 1] = (uint8_t)(cw >> 8); body[3 * i + 2] = (uint8_t)cw;
        }
        ob = 96;
    }
    const int nbits = lsf ? 240 : 144, P = lsf ? 61 : 12;
    unsigned mem = 0;
    int pi = 0;
    for (int i = 0; i < nbits + 4 && ob < 368; ++i) {
        const unsigned bit = i < nbits ? (rec[2 + (i >> 3)] >> (7 - (i & 7))) & 1u : 0u;
        mem = ((mem << 1) | bit) & 0x1Fu;
        const unsigned c[2] = {(unsigned)__popc(mem & 0x19u) & 1u, (unsigned)__popc(mem & 0x17u) & 1u};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const bool keep = lsf ? (pi & 3) != 2 : pi != 11;
            if (++pi >= P) pi = 0;
            if (keep && ob < 368) { if (c[k]) body[ob >> 3] |= (uint8_t)(0x80u >> (ob & 7)); ++ob; }
        }
    }
    for (unsigned i = 0; i < 368; ++i) {
        if ((body[i >> 3] >> (7 - (i & 7))) & 1u) {
            const unsigned d = (45u * i + 92u * i * i) % 368u;
            il[d >> 3] |= (uint8_t)(0x80u >> (d & 7));
        }
    }
    frames[f * 48] = lsf ? 0x55 : 0xFF;
    frames[f * 48 + 1] = lsf ? 0xF7 : 0x5D;
    for (int i = 0; i < 46; ++i) frames[f * 48 + 2 + i] = il[i] ^ c_m17_seq[i];
}
void launch_m17_encode(const uint8_t* records, size_t n, uint8_t* frames, hipStream_t s)
{
    if (!n) return;
    hipLaunchKernelGGL(k_m17_encode, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, records, n, frames);
}
void launch_m17_decode(const uint8_t* frames, size_t n, uint8_t* records, hipStream_t s)
{
    if (!n) return;
    hipLaunchKernelGGL(k_m17_decode, dim3((unsigned)((n + M17_TPB - 1) / M17_TPB)), dim3(M17_TPB), 0, s, frames, n, records);
}

}  // namespace qrl
