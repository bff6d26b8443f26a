// kernels_deframe.hip — gr_deframer_bb on the device (reference src/gr/gr_deframer_bb.cpp:83-185).
// One wave64 per stream.  While searching, the wave looks at 64 bit positions at once: a __ballot packs the 64 input
// bits into one word, every lane rebuilds the shift register as it would stand after ITS position (carry-in register
// shifted up, ballot word bit-reversed and shifted down), compares it with the sync words, and a second ballot picks
// the first match -- the position the serial loop of the reference would have stopped at.  While a frame is open the
// lanes copy its bits in parallel.  The register is cleared at the end of every frame exactly like the block does.
#include "devmath.hpp"
#include "engine.hpp"

namespace qrl {

__device__ __forceinline__ int deframer_find(int type, uint32_t reg, int& nbits)
{
    uint32_t temp = type != 2 ? (reg & 0xFFFFu) : (reg & 0xFFu);
    nbits = type == 2 ? 8 : 16;
    if (type == 2 && temp == 0xB5u) return (int)temp;
    if (temp == 0x89EDu || temp == 0xED89u || temp == 0x98DEu || temp == 0xED77u || temp == 0x8CC8u) return (int)temp;
    temp = reg & 0xFFFFFFu;
    if (temp == 0x4C8A2Bu) { if (type != 2) nbits = 24; return (int)temp; }
    return 0;
}

__global__ __launch_bounds__(64) void k_deframe(const DeframeParams P)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    uint32_t n = P.counts ? P.counts[(size_t)b * P.count_stride] : P.n;
    if (n > P.stride) n = (uint32_t)P.stride;   // a device-side count never reaches past the stream's own row
    const uint8_t* in = P.bits + (size_t)b * P.stride;
    uint8_t* out = P.out + (size_t)b * P.out_cap;
    DeframeState st = P.st[b];
    uint32_t i = 0, no = 0;
    while (i < n) {
        if (st.found) {
            const uint32_t take = min(n - i, P.buf_len - st.idx);
            for (uint32_t k = lane; k < take; k += 64)
                if (no + k < P.out_cap) out[no + k] = in[i + k] & 1u;
            no += take; i += take; st.idx += take;
            if (st.idx >= P.buf_len) { st.found = 0; st.reg = 0; st.idx = 0; }
        } else {
            const uint32_t blk = min(64u, n - i);
            const uint32_t bit = (uint32_t)lane < blk ? (in[i + lane] & 1u) : 0u;
            const unsigned long long m = __ballot(bit != 0);
            const uint32_t low = (uint32_t)(__brevll(m) >> (63 - lane));
            const uint32_t reg_l = ((lane + 1 < 32) ? (st.reg << (lane + 1)) : 0u) | low;
            int nb = 0;
            const int ft = (uint32_t)lane < blk ? deframer_find(P.type, reg_l, nb) : 0;
            const unsigned long long mm = __ballot(ft != 0);
            if (mm) {
                const int l0 = __ffsll((long long)mm) - 1;
                const int ft0 = __shfl(ft, l0, 64), nb0 = __shfl(nb, l0, 64);
                if (lane < nb0 && no + lane < P.out_cap) out[no + lane] = (uint8_t)((ft0 >> (nb0 - 1 - lane)) & 1);
                no += (uint32_t)nb0;
                st.found = 1; st.idx = 0; st.reg = __shfl(reg_l, l0, 64);
                i += (uint32_t)l0 + 1u;
            } else {
                st.reg = __shfl(reg_l, (int)blk - 1, 64);
                i += blk;
            }
        }
    }
    if (lane == 0) {
        P.st[b] = st;
        P.out_counts[b] = no < P.out_cap ? no : (uint32_t)P.out_cap;
    }
}

// ---- gr_modem::synchronize / findSync / packBytes on the device (reference src/gr_modem.cpp:1119-1282, 980-994).
// Same wave-per-stream search as k_deframe with the mode's sync-word class; the frame's bits are collected into a per-stream
// bit buffer (a frame may span calls), then packed MSB first by all lanes into one record
// { u32 frame_type, u32 nbytes, payload padded to a multiple of 4 } of the stream's output.
__device__ __forceinline__ uint32_t modem_find_sync(int cls, uint32_t reg)
{
    if (cls == 0) return (reg & 0xFFu) == 0xB5u ? 0xB5u : 0u;
    if (cls == 3) {   // M17 (gr_modem.cpp:1187-1210): the 16-bit LSF / stream words first, else the 32-bit EOT word
        if ((reg & 0xFFFFu) == 0x55F7u) return 0x55F7u;
        if ((reg & 0xFFFFu) == 0xFF5Du) return 0xFF5Du;
        return reg == 0x555D555Du ? 0x555D555Du : 0u;
    }
    const uint32_t t24 = reg & 0xFFFFFFu;
    if (cls == 2) {
        if ((reg & 0xFFFFu) == 0xED89u) return 0xED89u;
        if (t24 == 0x89EDAAu || t24 == 0xED77AAu || t24 == 0x98DEAAu || t24 == 0x8CC8DDu || t24 == 0x4C8A2Bu) return t24;
        return 0u;
    }
    if (t24 == 0xDE98AAu || t24 == 0x98DEAAu || t24 == 0x4C8A2Bu) return t24;
    return 0u;
}

__global__ __launch_bounds__(64) void k_framesync(const FrameSyncParams P)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    uint32_t n = P.counts ? P.counts[(size_t)b * P.count_stride] : P.n;
    if (n > P.stride) n = (uint32_t)P.stride;   // a device-side count never reaches past the stream's own row
    const uint8_t* in = P.bits + (size_t)b * P.stride;
    uint8_t* out = P.out + (size_t)b * P.out_cap;
    uint8_t* bitbuf = P.bitbuf + (size_t)b * P.bitbuf_stride;
    FrameSyncState st = P.st[b];
    uint32_t i = 0, no = 0, nframes = 0, collected = 0;
    // bits of the open frame: [0, carry) sit in bitbuf (written by EARLIER launches), the rest is in[fstart ...] of this call;
    // nothing written in this launch is read back in it (no reliance on L1 coherence between lanes)
    uint32_t carry = st.found ? st.idx : 0u, fstart = 0u;
    while (i < n) {
        if (st.found) {
            const bool adj = P.cls == 1 || P.cls == 2;   // (the 1k modes and M17 take bit_buf_len / frame_length as they are, :1147-1166)
            const bool voice = adj && st.ftype == 0xED89u;
            const uint32_t need = (adj && !voice) ? P.bit_buf_len - 8u : P.bit_buf_len;
            const uint32_t flen = voice ? P.frame_length + 1u : P.frame_length;
            const uint32_t take = min(n - i, need - st.idx);
            i += take; st.idx += take; collected += take;
            if (st.idx >= need) {
                const uint32_t padded = (flen + 3u) & ~3u;
                if (no + 8u + padded <= P.out_cap) {
                    if (lane == 0) { reinterpret_cast<uint32_t*>(out + no)[0] = st.ftype; reinterpret_cast<uint32_t*>(out + no)[1] = flen | (st.modem_sync << 16); }   // (_modem_sync <= 39)
                    for (uint32_t j = lane; j < padded; j += 64) {
                        uint32_t t = 0;
                        if (8u * j < need) {
#pragma unroll
                            for (int k = 0; k < 8; ++k) {
                                const uint32_t pb = 8u * j + k;
                                const uint32_t v = pb < carry ? bitbuf[pb] : in[fstart + (pb - carry)];
                                t = (t << 1) | (v & 1u);
                            }
                        }
                        out[no + 8u + j] = (uint8_t)t;
                    }
                    no += 8u + padded; ++nframes;
                }
                st.found = 0; st.reg = 0; st.idx = 0; carry = 0;
            }
        } else {
            const uint32_t blk = min(64u, n - i);
            const uint32_t bit = (uint32_t)lane < blk ? (in[i + lane] & 1u) : 0u;
            const unsigned long long m = __ballot(bit != 0);
            const uint32_t low = (uint32_t)(__brevll(m) >> (63 - lane));
            const uint32_t reg_l = ((lane + 1 < 32) ? (st.reg << (lane + 1)) : 0u) | low;
            const uint32_t ft = (uint32_t)lane < blk ? modem_find_sync(P.cls, reg_l) : 0u;
            const unsigned long long mm = __ballot(ft != 0);
            if (mm) {
                const int l0 = __ffsll((long long)mm) - 1;
                st.ftype = __shfl(ft, l0, 64);
                st.found = 1; st.idx = 0; st.reg = __shfl(reg_l, l0, 64);
                // _modem_sync: -1 (floor 0) for each of the l0 bits searched before the word completed, then +8 below 32
                st.modem_sync = st.modem_sync > (uint32_t)l0 ? st.modem_sync - (uint32_t)l0 : 0u;
                if (st.modem_sync < 32u) st.modem_sync += 8u;
                i += (uint32_t)l0 + 1u;
                fstart = i; carry = 0;
            } else {
                st.reg = __shfl(reg_l, (int)blk - 1, 64);
                st.modem_sync = st.modem_sync > blk ? st.modem_sync - blk : 0u;
                i += blk;
            }
        }
    }
    if (st.found) {   // the call ends inside a frame: keep this call's part of it for the next launch
        for (uint32_t k = lane; fstart + k < n; k += 64) bitbuf[carry + k] = in[fstart + k] & 1u;
    }
    if (lane == 0) {
        P.st[b] = st;
        P.out_counts[2 * b] = no;
        P.out_counts[2 * b + 1] = nframes;
        if (P.activity) P.activity[b] = collected;   // > 0 <=> gr_modem::synchronize returns data_to_process = true for these bits (src/gr_modem.cpp:1121-1175)
    }
}
void launch_framesync(const FrameSyncParams& p, int batch, hipStream_t s)
{
    hipLaunchKernelGGL(k_framesync, dim3(batch), dim3(64), 0, s, p);
}

void launch_deframe(const DeframeParams& p, int batch, hipStream_t s)
{
    hipLaunchKernelGGL(k_deframe, dim3(batch), dim3(64), 0, s, p);
}

}  // namespace qrl
