// kernels_tx.hip — TX side of the path: gr_mod_qpsk (reference src/gr/gr_mod_qpsk.cpp:56-89, instance
// make_gr_mod_qpsk(4, 1000000, 1700, 160000) src/gr/gr_mod_base.cpp:175):
//   packed_to_unpacked_bb(1, MSB) -> scrambler_bb(0x8A, 0x7F, 7) -> cc_encoder(K=7, {109, 79}) -> pack_k_bits(2)
//   -> map_bb{0,1,3,2} -> diff_encoder_bb(4) -> chunks_to_symbols_bc -> rational_resampler_ccf(sps, 1, RRC)
//   -> multiply_const_cc(0.6) -> multiply_const_cc(bb_gain)
//
//  k_tx_qpsk_bits : ONE WAVE PER STREAM turns the bytes of a call into differential symbol indices.
//     The additive scrambler is a GF(2)-linear recurrence, so the 64 lanes each take a contiguous slice:
//     pass 1 runs the slice from a ZERO register (gives the slice's contribution to the final state), a
//     63-step lane chain composes it with T^L (the L-step zero-input transition, 8 column masks from the
//     host) to get every lane's TRUE start register, pass 2 reruns the slice for real.  The encoder is
//     feed-forward (6-bit halo) and the differential encoder is a prefix sum mod 4 (wave scan).
//  k_tx_interp    : polyphase interpolating FIR, one thread per output sample, one fmaf chain per output
//     (j ascending, as oracle orc_resamp_ccf), constellation looked up from the symbol index.
#include "devmath.hpp"
#include "engine.hpp"

namespace qrl {

__device__ __forceinline__ uint32_t lfsr_step(uint32_t sr, uint32_t in)   // scrambler_bb(0x8A, seed, 7)
{
    const uint32_t nb = (__builtin_popcount(sr & 0x8Au) & 1u) ^ (in & 1u);
    return (sr >> 1) | (nb << 7);
}
__device__ __forceinline__ uint32_t gf2_apply(const uint8_t* cols, uint32_t v)
{
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) r ^= ((v >> k) & 1u) ? cols[k] : 0u;
    return r;
}

// Round 4: word-parallel where the algebra allows it.  The lane's slice of L input bits is held as 32-bit words (bit j of word w =
// input bit lane L + 32 w + j, i.e. the bytes bit-reversed); the true start register of every lane comes from a log-step scan over the
// lanes (the slice maps are affine over GF(2): after step d a lane holds the composition of its 2^(d+1) predecessors, combined with the
// zero-input transition T^(L 2^d) -- six 8 x 8 column-mask matrices from the host) instead of a 63-step chain; the two coded bits of
// 32 consecutive input bits are XORs of shifted copies of the scrambled word (c0 = s ^ s>>2 ^ s>>3 ^ s>>5 ^ s>>6 for 109, c1 = s ^ s>>1 ^
// s>>2 ^ s>>3 ^ s>>6 for 79, on the 38-bit window prev6 | word), map{0,1,3,2}[2 c0 + c1] = 2 c0 + (c0 ^ c1), and a slice's symbol sum is
// two popcounts.  What stays bit-serial: the scrambler itself (a recursion with input feedback: two register-only passes per slice)
// and the running differential symbol.  Symbols leave four to a dword.  (tools: the lane algorithm was checked against a serial model
// in Python before it was written here; tests/test_gpu_tx.py has the bit-exact and chunk-invariance cases, sizes 1 .. 8192 bytes.)
__global__ __launch_bounds__(64) void k_tx_qpsk_bits(const TxBitsParams P)
{
    extern __shared__ __align__(16) unsigned char tx_smem[];
    uint32_t* sbits = reinterpret_cast<uint32_t*>(tx_smem);     // scrambled bits, word lane * nw + w, LSB = earliest
    const int b = blockIdx.x, lane = threadIdx.x;
    const uint8_t* in = P.bytes + (size_t)b * P.stride;
    TxState st = P.st[b];
    const uint32_t nbits = P.nbytes * 8u;
    const uint32_t L = P.L, nw = L >> 5;                        // bits / words per lane
    const uint32_t hi = min(nbits, (uint32_t)(lane + 1) * L);                  // this lane's slice = input bits [lane L, hi)
    const bool aligned4 = (reinterpret_cast<uintptr_t>(in) & 3u) == 0;
    auto in_word = [&](uint32_t w) -> uint32_t {                 // bit j = input bit lane L + 32 w + j (packed_to_unpacked MSB first), 0 behind the end
        const uint32_t byte0 = ((uint32_t)lane * L >> 3) + 4u * w;
        if (byte0 >= P.nbytes) return 0u;
        uint32_t d = 0;
        if (aligned4 && byte0 + 4u <= P.nbytes) d = *reinterpret_cast<const uint32_t*>(in + byte0);
        else
            for (uint32_t k = 0; k < 4u; ++k) if (byte0 + k < P.nbytes) d |= (uint32_t)in[byte0 + k] << (8u * k);
        return __builtin_bswap32(__builtin_bitreverse32(d));
    };
    auto valid_bits = [&](uint32_t w) -> uint32_t {              // how many bits of word w lie inside [lo, hi)
        const uint32_t i0 = (uint32_t)lane * L + 32u * w;
        return hi > i0 ? min(32u, hi - i0) : 0u;
    };

    // pass 1: the slice from a ZERO register = its contribution to the register behind it
    uint32_t sf = 0;
    for (uint32_t w = 0; w < nw; ++w) {
        const uint32_t x = in_word(w), nv = valid_bits(w);
        for (uint32_t j = 0; j < nv; ++j) sf = lfsr_step(sf, x >> j);
    }
    // registers behind every lane's slice by an inclusive scan: c[l] = T^L c[l-1] ^ sf[l], c[-1] = st.sr.  Only FULL slices feed later
    // lanes that have bits (L = the slice length of every lane but the last one with bits), so T^(L 2^d) is the right power throughout.
    uint32_t c = sf;
    if (lane == 0) c ^= gf2_apply(P.tl_pow[0], st.sr);
#pragma unroll
    for (int d = 0; d < 6; ++d) {
        const uint32_t t = __shfl_up(c, 1u << d, 64);
        if (lane >= (1 << d)) c ^= gf2_apply(P.tl_pow[d], t);
    }
    const uint32_t before = __shfl_up(c, 1, 64);
    // pass 2: the real scrambler; output bit = sr & 1 BEFORE the step
    uint32_t sr = lane == 0 ? st.sr : before;
    for (uint32_t w = 0; w < nw; ++w) {
        const uint32_t x = in_word(w), nv = valid_bits(w);
        uint32_t word = 0;
        for (uint32_t j = 0; j < nv; ++j) {
            word |= (sr & 1u) << j;
            sr = lfsr_step(sr, x >> j);
        }
        sbits[(uint32_t)lane * nw + w] = word;
    }
    // register after the whole call = register of the last lane that had bits
    const uint32_t last_lane = nbits ? (nbits - 1) / L : 0;
    const uint32_t sr_end = __shfl(sr, (int)last_lane, 64);
    __syncthreads();

    // coded bits of word w of this lane: c0 / c1 bit t belong to input bit lane L + 32 w + t
    auto coded = [&](uint32_t w, uint32_t& c0, uint32_t& c1) {
        const uint32_t gi = (uint32_t)lane * nw + w;
        // the 6 scrambled bits in front of the word, bit j = scrambled bit (first bit of the word) - 6 + j; in front of the call: st.enc
        // (bit k = scrambled bit -1 - k)
        const uint32_t prev6 = gi == 0 ? (__builtin_bitreverse32(st.enc & 63u) >> 26) : (sbits[gi - 1] >> 26);
        const uint64_t T = ((uint64_t)sbits[gi] << 6) | prev6;
        c0 = (uint32_t)((T >> 6) ^ (T >> 4) ^ (T >> 3) ^ (T >> 1) ^ T);      // taps of 109: reg bits 0, 2, 3, 5, 6 (bit k = scrambled bit i - k)
        c1 = (uint32_t)((T >> 6) ^ (T >> 5) ^ (T >> 4) ^ (T >> 3) ^ T);      // taps of 79:  reg bits 0, 1, 2, 3, 6
    };
    uint8_t* ring = P.sym.p + (size_t)b * (P.sym.mask + 1u);
    auto enc_state = [&]() -> uint32_t {                          // bit k = scrambled bit nbits - 1 - k (nbits >= 8: all of this call)
        uint32_t enc = 0;
        for (int k = 0; k < 6; ++k) { const uint32_t i = nbits - 1u - (uint32_t)k; enc |= ((sbits[i >> 5] >> (i & 31u)) & 1u) << k; }
        return enc;
    };
    if (P.mode == 2 || P.mode == 1) {
        // 4FSK (gr_mod_4fsk.cpp:96-101): pack_k_bits(2) -> map{0,1,3,2}, one symbol index per input bit, no differential coding;
        // FSK family: the two coded bits of every input bit go to the ring as they are (chunks_to_symbols later)
        for (uint32_t w = 0; w < nw; ++w) {
            const uint32_t nv = valid_bits(w);
            if (!nv) break;
            uint32_t c0, c1;
            coded(w, c0, c1);
            const uint32_t g = c0 ^ c1, i0 = (uint32_t)lane * L + 32u * w;
            if (P.mode == 2) {
                for (uint32_t t = 0; t < nv; t += 4) {            // (nv is a multiple of 8: whole bytes)
                    uint32_t pk = 0;
#pragma unroll
                    for (uint32_t u = 0; u < 4; ++u) pk |= ((((c0 >> (t + u)) & 1u) << 1) | ((g >> (t + u)) & 1u)) << (8u * u);
                    *reinterpret_cast<uint32_t*>(ring + ((uint32_t)(P.s0 + i0 + t) & P.sym.mask)) = pk;
                }
            } else {
                for (uint32_t t = 0; t < nv; t += 2) {
                    const uint32_t pk = ((c0 >> t) & 1u) | (((c1 >> t) & 1u) << 8) | (((c0 >> (t + 1)) & 1u) << 16) | (((c1 >> (t + 1)) & 1u) << 24);
                    *reinterpret_cast<uint32_t*>(ring + ((uint32_t)(P.s0 + 2ull * (i0 + t)) & P.sym.mask)) = pk;
                }
            }
        }
        if (lane == 0 && nbits) { st.sr = sr_end; st.enc = enc_state(); P.st[b] = st; }
        return;
    }
    // QPSK: diff_encoder_bb(4): y[n] = (x[n] + y[n-1]) mod 4, x = map{0,1,3,2}[2 c0 + c1] = 2 c0 + (c0 ^ c1)
    uint32_t local = 0;                                          // sum of the slice's symbols
    for (uint32_t w = 0; w < nw; ++w) {
        const uint32_t nv = valid_bits(w);
        if (!nv) break;
        uint32_t c0, c1;
        coded(w, c0, c1);
        const uint32_t vm = nv >= 32u ? 0xffffffffu : ((1u << nv) - 1u);
        local += 2u * (uint32_t)__builtin_popcount(c0 & vm) + (uint32_t)__builtin_popcount((c0 ^ c1) & vm);
    }
    local &= 3u;
    uint32_t incl = local;                                       // exclusive wave scan of the slice sums
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, 64);
        if (lane >= off) incl = (incl + o) & 3u;
    }
    uint32_t run = (st.prev + incl - local) & 3u;                // symbol before this lane's slice
    for (uint32_t w = 0; w < nw; ++w) {
        const uint32_t nv = valid_bits(w);
        if (!nv) break;
        uint32_t c0, c1;
        coded(w, c0, c1);
        const uint32_t g = c0 ^ c1, i0 = (uint32_t)lane * L + 32u * w;
        for (uint32_t t = 0; t < nv; t += 4) {
            uint32_t pk = 0;
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) {
                run = (run + ((((c0 >> (t + u)) & 1u) << 1) | ((g >> (t + u)) & 1u))) & 3u;
                pk |= run << (8u * u);
            }
            *reinterpret_cast<uint32_t*>(ring + ((uint32_t)(P.s0 + i0 + t) & P.sym.mask)) = pk;
        }
    }
    const uint32_t prev_end = __shfl(run, (int)last_lane, 64);
    if (lane == 0 && nbits) { st.sr = sr_end; st.enc = enc_state(); st.prev = prev_end; P.st[b] = st; }
}

void launch_tx_qpsk_bits(const TxBitsParams& p, int batch, hipStream_t s)
{
    if (!p.nbytes) return;
    const size_t lds = (size_t)64 * (p.L >> 5) * 4 + 16;
    if (lds > 64 * 1024 && dyn_lds_limit(reinterpret_cast<const void*>(k_tx_qpsk_bits), (int)lds) != hipSuccess) return;
    hipLaunchKernelGGL(k_tx_qpsk_bits, dim3(batch), dim3(64), lds, s, p);
}

__global__ __launch_bounds__(256) void k_tx_interp(const TxInterpParams P)
{
    __shared__ float taps_s[256];
    taps_s[threadIdx.x] = (int)threadIdx.x < P.nt ? P.taps[threadIdx.x] : 0.f;
    __syncthreads();
    const float* taps = P.nt <= 256 ? taps_s : P.taps;   // BPSK: 11 x sps taps (sps = 500 / 250) stay in global memory / L2
    const int b = blockIdx.y;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= P.count) return;
    const uint64_t n = P.n0 + t;                       // absolute output sample
    const int I = P.interp;
    const uint64_t c = n / (uint64_t)I;
    const int ph = (int)(n - c * (uint64_t)I);
    const uint8_t* ring = P.sym.p + (size_t)b * (P.sym.mask + 1u);
    float ar = 0.f, ai = 0.f;
    for (int j = 0; ph + j * I < P.nt; ++j) {
        if ((uint64_t)j > c) break;                    // symbols before the stream start are zero
        const float h = taps[ph + j * I];
        const float2 x = P.table[ring[(uint32_t)(c - j) & P.sym.mask] & 3u];
        ar = fmaf(h, x.x, ar);
        ai = fmaf(h, x.y, ai);
    }
    ar *= P.amp; ai *= P.amp;                          // multiply_const_cc(0.6)
    ar *= P.bb_gain; ai *= P.bb_gain;                  // multiply_const_cc(bb_gain)
    P.out[(size_t)b * P.out_stride + t] = make_float2(ar, ai);
}

// Small interpolation factors (QPSK-250k: 4 samples per symbol, 61 taps): one thread per SYMBOL writes its I output samples.
// The workgroup stages the constellation points of its 256 symbols + J - 1 predecessors in LDS once, instead of one 64-bit division,
// ~16 byte loads and ~16 table loads per output sample (C5: 4.7 ms for 268 M samples, 0.45 TB/s of stores); the taps are scalar
// operands (round 4: they were LDS broadcasts, 64 ds_reads per thread beside 64 packed fmas).
// Same fmaf chain per output (j ascending); the zero-padded taps / the zero symbols in front of the stream add +0.
template <int I, int J>
__global__ __launch_bounds__(256) void k_tx_interp_sym(const TxInterpParams P)
{
    __shared__ float2 xs[256 + J];
    const float* taps = P.taps;                                   // wave uniform, compile-time offsets: scalar loads, SGPR operands of the fmas (the table is followed by 64 zeros, tx.cpp upload)
    const int b = blockIdx.y, tid = threadIdx.x;
    const uint64_t cs = P.n0 / (uint64_t)I;                       // first symbol of this call (n0 is a multiple of I)
    const uint32_t nsym = P.count / (uint32_t)I;
    const uint32_t s0 = blockIdx.x * 256u;
    const uint8_t* ring = P.sym.p + (size_t)b * (P.sym.mask + 1u);
    for (int i = tid; i < 256 + J - 1; i += 256) {                // xs[i] = symbol cs + s0 - (J - 1) + i
        const int64_t c = (int64_t)(cs + s0) - (J - 1) + i;
        xs[i] = c >= 0 ? P.table[ring[(uint32_t)c & P.sym.mask] & 3u] : make_float2(0.f, 0.f);
    }
    __syncthreads();
    __shared__ float2 ys[I][256 + 8];                             // outputs of the workgroup, [phase][symbol] (row pitch 264: the four rows start 16 banks apart)
    if (s0 + tid < nsym) {
        float2 xv[J];                                             // the J symbols under the filter, read once for the I phases
#pragma unroll
        for (int j = 0; j < J; ++j) xv[j] = xs[J - 1 + tid - j];
#pragma unroll
        for (int ph = 0; ph < I; ++ph) {
            float ar = 0.f, ai = 0.f;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const float h = taps[ph + j * I];
                ar = fmaf(h, xv[j].x, ar);
                ai = fmaf(h, xv[j].y, ai);
            }
            ar *= P.amp; ai *= P.amp;                             // multiply_const_cc(0.6)
            ar *= P.bb_gain; ai *= P.bb_gain;                     // multiply_const_cc(bb_gain)
            ys[ph][tid] = make_float2(ar, ai);
        }
    }
    __syncthreads();
    // a thread owns I consecutive output samples: stored from its registers, every instruction would scatter 8-byte pieces at a
    // 32-byte stride.  Through the LDS they leave in order, 512 contiguous bytes per store instruction (k_dec2_fir: same finding).
    float2* o = P.out + (size_t)b * P.out_stride + (size_t)s0 * I;
    const uint32_t nout = (nsym > s0 ? (nsym - s0 < 256u ? nsym - s0 : 256u) : 0u) * (uint32_t)I;
#pragma unroll
    for (int k = 0; k < I; ++k) {
        const uint32_t idx = (uint32_t)tid + 256u * k;            // output s0 * I + idx = (symbol idx / I, phase idx % I)
        if (idx < nout) o[idx] = ys[idx % I][idx / I];
    }
}

void launch_tx_interp(const TxInterpParams& p, int batch, hipStream_t s)
{
    if (!p.count) return;
    if (p.interp == 4 && p.nt <= 64 && p.n0 % 4 == 0 && p.count % 4 == 0) {
        hipLaunchKernelGGL((k_tx_interp_sym<4, 16>), dim3((p.count / 4 + 255) / 256, batch), dim3(256), 0, s, p);
        return;
    }
    hipLaunchKernelGGL(k_tx_interp, dim3((p.count + 255) / 256, batch), dim3(256), 0, s, p);
}

// ---- FSK family (gr_mod_2fsk.cpp:63-99, gr_mod_gmsk.cpp:68-95):
//   chunks_to_symbols_bf{-1, 1} -> repeat(sps) | rational_resampler_fff(sps, 1, RRC | gaussian) -> frequency_modulator_fc(k)
//   -> multiply_const(amplif) -> rational_resampler_ccf(I2, 1, low_pass(I2, samp_rate, fw, fw))
// k_tx_shape : thread per sample at the symbol-interpolated rate: +-1 repeated, or the polyphase shaping FIR.
// k_tx_fm    : the FM phase accumulator wraps with fmodf after every sample, so it is a serial float recurrence per
//              stream: one lane per stream, windows staged through LDS; (cos, sin) from the deterministic polynomial.
// k_tx_interp_c : second interpolating FIR on the complex FM output (I2 = 1: a plain FIR).
__global__ __launch_bounds__(256) void k_tx_shape(const TxShapeParams P)
{
    __shared__ float taps[1536];
    for (int k = threadIdx.x; k < P.nt && k < 1536; k += 256) taps[k] = P.taps[k];
    __syncthreads();
    const int b = blockIdx.y;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= P.count) return;
    const uint64_t n = P.n0 + t;
    const uint64_t c = n / (uint64_t)P.sps;
    const int ph = (int)(n - c * (uint64_t)P.sps);
    const uint8_t* ring = P.sym.p + (size_t)b * (P.sym.mask + 1u);
    // chunks_to_symbols_bf: {-1, 1} (2FSK / GMSK) or {-1.5, -0.5, 0.5, 1.5} (4FSK, gr_mod_4fsk.cpp:33-38)
    auto level = [&](uint8_t v) -> float { return P.levels == 4 ? (float)(v & 3) - 1.5f : (v ? 1.0f : -1.0f); };
    float a;
    if (P.nt == 0) a = level(ring[(uint32_t)c & P.sym.mask]);                  // blocks::repeat
    else {
        a = 0.f;
        for (int j = 0; ph + j * P.sps < P.nt; ++j) {
            if ((uint64_t)j > c) break;
            a = fmaf(taps[ph + j * P.sps], level(ring[(uint32_t)(c - j) & P.sym.mask]), a);
        }
        if (P.scale != 0.0f) a = a * P.scale;                                  // _scale_pulses (4FSK FM, gr_mod_4fsk.cpp:88)
    }
    P.out.p[(size_t)b * (P.out.mask + 1u) + ((uint32_t)n & P.out.mask)] = a;
}
void launch_tx_shape(const TxShapeParams& p, int batch, hipStream_t s)
{
    if (!p.count) return;
    hipLaunchKernelGGL(k_tx_shape, dim3((p.count + 255) / 256, batch), dim3(256), 0, s, p);
}

constexpr int TXF_CH = 96;
__global__ __launch_bounds__(64) void k_tx_fm(const TxFmParams P, int batch)
{
    __shared__ float win[64][TXF_CH + 1];
    __shared__ float2 wout[64][TXF_CH / 2 + 1];   // flushed in two halves to stay inside 64 KB static LDS
    const int lane = threadIdx.x, b0 = blockIdx.x * 64, b = b0 + lane;
    const bool active = b < batch;
    const int nstreams = min(64, batch - b0);
    float phase = active ? P.phase[b] : 0.f;
    const float F_PI = 3.14159265358979323846f;
    for (uint32_t c0 = 0; c0 < P.count; c0 += TXF_CH) {
        const int len = min((uint32_t)TXF_CH, P.count - c0);
        __syncthreads();
        for (int s = 0; s < nstreams; ++s)
            for (int k = lane; k < len; k += 64)
                win[s][k] = P.in.p[(size_t)(b0 + s) * (P.in.mask + 1u) + ((uint32_t)(P.n0 + c0 + k) & P.in.mask)];
        __syncthreads();
        for (int half = 0; half < 2; ++half) {
            const int k0 = half * (TXF_CH / 2), k1 = min(len, k0 + TXF_CH / 2);
            if (active) {
                for (int k = k0; k < k1; ++k) {
                    phase = phase + P.k * win[lane][k];
                    phase = fmodf(phase + F_PI, 2.0f * F_PI) - F_PI;
                    const float2 cs = sincos_rad(phase);
                    wout[lane][k - k0] = make_float2(cs.x * P.amp, cs.y * P.amp);
                }
            }
            __syncthreads();
            for (int s = 0; s < nstreams; ++s)
                for (int k = k0 + lane; k < k1; k += 64)
                    P.out.p[(size_t)(b0 + s) * (P.out.mask + 1u) + ((uint32_t)(P.n0 + c0 + k) & P.out.mask)] = wout[s][k - k0];
            __syncthreads();
        }
    }
    if (active) P.phase[b] = phase;
}
void launch_tx_fm(const TxFmParams& p, int batch, hipStream_t s)
{
    if (!p.count) return;
    hipLaunchKernelGGL(k_tx_fm, dim3((batch + 63) / 64), dim3(64), 0, s, p, batch);
}

__global__ __launch_bounds__(256) void k_tx_interp_c(const TxInterpCParams P)
{
    __shared__ float taps_s[2048];
    const bool in_lds = P.nt <= 2048;      // longer filters (gr_mod_base back end at high device rates) read taps through L1/L2
    if (in_lds) for (int k = threadIdx.x; k < P.nt; k += 256) taps_s[k] = P.taps[k];
    __syncthreads();
    const float* taps = in_lds ? taps_s : P.taps;
    const int b = blockIdx.y;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= P.count) return;
    const uint64_t n = P.n0 + t;
    const uint64_t u = n * (uint64_t)(P.decim > 1 ? P.decim : 1);         // rational_resampler_ccf(interp, decim): output n sits at u / interp
    const uint64_t c = u / (uint64_t)P.interp;
    const int ph = (int)(u - c * (uint64_t)P.interp);
    const float2* ring = P.in.p + (size_t)b * (P.in.mask + 1u);
    float ar = 0.f, ai = 0.f;
    for (int j = 0; ph + j * P.interp < P.nt; ++j) {
        if ((uint64_t)j > c) break;
        const float h = taps[ph + j * P.interp];
        const float2 x = ring[(uint32_t)(c - j) & P.in.mask];
        ar = fmaf(h, x.x, ar);
        ai = fmaf(h, x.y, ai);
    }
    if (P.out_ring.p) P.out_ring.p[(size_t)b * (P.out_ring.mask + 1u) + ((uint32_t)n & P.out_ring.mask)] = make_float2(ar, ai);
    else P.out[(size_t)b * P.out_stride + t] = make_float2(ar, ai);
}
// gr_mod_m17 (reference src/gr/gr_mod_m17.cpp:47-58,77-80): packed_to_unpacked(1, MSB) -> pack_k_bits(2) -> map{2, 3, 1, 0}: four symbol
// indices per byte, no scrambler / FEC (the M17 frame encoder has already done that)
__global__ __launch_bounds__(256) void k_tx_raw_dibits(const uint8_t* __restrict__ bytes, size_t stride, uint32_t nbytes, RingB sym, uint64_t s0)
{
    const int b = blockIdx.y;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= nbytes * 4u) return;
    const uint32_t v = (bytes[(size_t)b * stride + (t >> 2)] >> (6u - 2u * (t & 3u))) & 3u;
    const uint32_t map = (0x1Eu >> (2u * v)) & 3u;                       // {2, 3, 1, 0}
    sym.p[(size_t)b * (sym.mask + 1u) + ((uint32_t)(s0 + t) & sym.mask)] = (uint8_t)map;
}
void launch_tx_raw_dibits(const uint8_t* bytes, size_t stride, uint32_t nbytes, RingB sym, uint64_t s0, int batch, hipStream_t s)
{
    if (!nbytes) return;
    hipLaunchKernelGGL(k_tx_raw_dibits, dim3((nbytes * 4u + 255) / 256, batch), dim3(256), 0, s, bytes, stride, nbytes, sym, s0);
}
// gr_mod_dsss (reference src/gr/gr_mod_dsss.cpp:27-92): dsss_encoder_bb (src/gr/dsss_encoder_bb_impl.cc:78-92) spreads every coded bit
// by the Barker-13 code (bit 0: the code, bit 1: its complement), 13 chips per ring item of the coded-bit ring
__global__ __launch_bounds__(256) void k_tx_spread(RingB coded, RingB chips, uint64_t c0, uint32_t count)
{
    const int b = blockIdx.y;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= count) return;
    const uint64_t ch = c0 * 13ull + t;                              // absolute chip index
    const uint32_t bit = coded.p[(size_t)b * (coded.mask + 1u) + ((uint32_t)(ch / 13ull) & coded.mask)] & 1u;
    const uint32_t code = (0x1F35u >> (12u - (uint32_t)(ch % 13ull))) & 1u;   // 1 1 1 1 1 0 0 1 1 0 1 0 1, first chip = MSB
    chips.p[(size_t)b * (chips.mask + 1u) + ((uint32_t)ch & chips.mask)] = (uint8_t)(bit ? code ^ 1u : code);
}
void launch_tx_spread(RingB coded, RingB chips, uint64_t c0, uint32_t ncoded, int batch, hipStream_t s)
{
    if (!ncoded) return;
    hipLaunchKernelGGL(k_tx_spread, dim3((ncoded * 13u + 255) / 256, batch), dim3(256), 0, s, coded, chips, c0, ncoded * 13u);
}
// float ring -> complex ring with a gain: (x g, 0)  (multiply_const_cc on a stream whose imaginary part is zero)
__global__ __launch_bounds__(256) void k_tx_f2c(RingF in, RingC out, uint64_t n0, uint32_t count, float g)
{
    const int b = blockIdx.y;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= count) return;
    const uint32_t n = (uint32_t)(n0 + t);
    out.p[(size_t)b * (out.mask + 1u) + (n & out.mask)] = make_float2(in.p[(size_t)b * (in.mask + 1u) + (n & in.mask)] * g, 0.0f);
}
void launch_tx_f2c(RingF in, RingC out, uint64_t n0, uint32_t count, float g, int batch, hipStream_t s)
{
    if (!count) return;
    hipLaunchKernelGGL(k_tx_f2c, dim3((count + 255) / 256, batch), dim3(256), 0, s, in, out, n0, count, g);
}
void launch_tx_interp_c(const TxInterpCParams& p, int batch, hipStream_t s)
{
    if (!p.count) return;
    hipLaunchKernelGGL(k_tx_interp_c, dim3((p.count + 255) / 256, batch), dim3(256), 0, s, p);
}

// ---- gr_mod_base back end (reference src/gr/gr_mod_base.cpp:38,249-258): rotator_cc(2 pi offset / 1e6) at 1 Msps, then
// rational_resampler_ccf(fs/1e6, 1, low_pass(I, fs, 480k, 20k, BH)) (k_tx_interp_c).  Exact 2^-64-turn NCO as on RX.
__global__ __launch_bounds__(256) void k_tx_rot(const TxRotParams P)
{
    const int b = blockIdx.y;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= P.count) return;
    const uint64_t n = P.n0 + t;
    const float2 x = P.in[(size_t)b * P.in_stride + t];
    const uint64_t kk = n - P.rot_nbase;
    const float2 hi = sincos_turn(P.rot_acc + ((kk >> 9) << 9) * P.rot_inc);
    const float2 y = cmul_fma(x, cmul_fma(hi, P.rot_lo[(uint32_t)kk & 511u]));
    if (P.out_ring.p) P.out_ring.p[(size_t)b * (P.out_ring.mask + 1u) + ((uint32_t)n & P.out_ring.mask)] = y;
    else P.out[(size_t)b * P.out_stride + t] = y;
}
void launch_tx_rot(const TxRotParams& p, int batch, hipStream_t s)
{
    if (!p.count) return;
    hipLaunchKernelGGL(k_tx_rot, dim3((p.count + 255) / 256, batch), dim3(256), 0, s, p);
}

}  // namespace qrl
