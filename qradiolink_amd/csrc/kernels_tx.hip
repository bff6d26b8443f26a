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
__device__ __forceinline__ uint32_t gf2_apply(const uint8_t (&cols)[8], uint32_t v)
{
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) r ^= ((v >> k) & 1u) ? cols[k] : 0u;
    return r;
}

__global__ __launch_bounds__(64) void k_tx_qpsk_bits(const TxBitsParams P)
{
    extern __shared__ __align__(16) unsigned char tx_smem[];
    uint32_t* sbits = reinterpret_cast<uint32_t*>(tx_smem);     // scrambled bits of this call, packed LSB = earliest
    const int b = blockIdx.x, lane = threadIdx.x;
    const uint8_t* in = P.bytes + (size_t)b * P.stride;
    TxState st = P.st[b];
    const uint32_t nbits = P.nbytes * 8u;
    const uint32_t L = P.L;                                     // bits per lane, multiple of 32
    const uint32_t lo = min(nbits, lane * L), hi = min(nbits, (lane + 1) * L);
    auto in_bit = [&](uint32_t i) { return (uint32_t)(in[i >> 3] >> (7u - (i & 7u))) & 1u; };   // packed_to_unpacked MSB first

    // pass 1: slice from a zero register
    uint32_t sf = 0;
    for (uint32_t i = lo; i < hi; ++i) sf = lfsr_step(sf, in_bit(i));
    // true start register of every lane: init[l] = T^len(l-1) init[l-1] ^ sf[l-1]; only full slices use T^L,
    // a ragged or empty slice (lanes past the end) never feeds a later lane that has bits
    uint8_t cols[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) cols[k] = P.tl_cols[k];
    uint32_t init = st.sr;
    uint32_t mine = st.sr;
    for (int l = 1; l < 64; ++l) {
        const uint32_t sfp = __shfl(sf, l - 1, 64);
        init = gf2_apply(cols, init) ^ sfp;
        if (lane == l) mine = init;
    }
    // pass 2: the real scrambler; output bit = sr & 1 BEFORE the step
    uint32_t sr = mine, word = 0;
    for (uint32_t i = lo; i < hi; ++i) {
        word |= (sr & 1u) << (i & 31u);
        sr = lfsr_step(sr, in_bit(i));
        if ((i & 31u) == 31u || i + 1 == hi) { sbits[i >> 5] = word; word = 0; }
    }
    // register after the whole call = register of the last lane that had bits
    const uint32_t last_lane = nbits ? (nbits - 1) / L : 0;
    const uint32_t sr_end = __shfl(sr, (int)last_lane, 64);
    __syncthreads();

    // encoder + map: one symbol per input bit.  Scrambled bit i, i < 0 comes from the previous call (st.enc).
    auto sbit = [&](int64_t i) -> uint32_t {
        if (i >= 0) return (sbits[i >> 5] >> (i & 31)) & 1u;
        return (st.enc >> (uint32_t)(-i - 1)) & 1u;              // st.enc bit k = scrambled bit (-1 - k)
    };
    const uint32_t map4[4] = {0u, 1u, 3u, 2u};
    uint32_t local = 0;
    for (uint32_t i = lo; i < hi; ++i) {
        uint32_t reg = 0;                                        // bit k = scrambled bit i - k (cc_encoder shift register)
#pragma unroll
        for (int k = 0; k < 7; ++k) reg |= sbit((int64_t)i - k) << k;
        const uint32_t c0 = __builtin_popcount(reg & 109u) & 1u, c1 = __builtin_popcount(reg & 79u) & 1u;
        local = (local + map4[(c0 << 1) | c1]) & 3u;
    }
    if (P.mode == 2) {   // 4FSK (gr_mod_4fsk.cpp:96-101): pack_k_bits(2) -> map{0,1,3,2}, one symbol index per input bit, no differential coding
        uint8_t* ring2 = P.sym.p + (size_t)b * (P.sym.mask + 1u);
        for (uint32_t i = lo; i < hi; ++i) {
            uint32_t reg = 0;
#pragma unroll
            for (int k = 0; k < 7; ++k) reg |= sbit((int64_t)i - k) << k;
            const uint32_t c0 = __builtin_popcount(reg & 109u) & 1u, c1 = __builtin_popcount(reg & 79u) & 1u;
            ring2[(uint32_t)(P.s0 + i) & P.sym.mask] = (uint8_t)map4[(c0 << 1) | c1];
        }
        if (lane == 0 && nbits) {
            uint32_t enc = 0;
            for (int k = 0; k < 6; ++k) enc |= sbit((int64_t)nbits - 1 - k) << k;
            st.sr = sr_end; st.enc = enc;
            P.st[b] = st;
        }
        return;
    }
    if (P.mode == 1) {   // FSK family: the two coded bits of every input bit go to the ring as they are (chunks_to_symbols later)
        uint8_t* ring1 = P.sym.p + (size_t)b * (P.sym.mask + 1u);
        for (uint32_t i = lo; i < hi; ++i) {
            uint32_t reg = 0;
#pragma unroll
            for (int k = 0; k < 7; ++k) reg |= sbit((int64_t)i - k) << k;
            ring1[(uint32_t)(P.s0 + 2ull * i) & P.sym.mask] = (uint8_t)(__builtin_popcount(reg & 109u) & 1u);
            ring1[(uint32_t)(P.s0 + 2ull * i + 1) & P.sym.mask] = (uint8_t)(__builtin_popcount(reg & 79u) & 1u);
        }
        if (lane == 0 && nbits) {
            uint32_t enc = 0;
            for (int k = 0; k < 6; ++k) enc |= sbit((int64_t)nbits - 1 - k) << k;
            st.sr = sr_end; st.enc = enc;
            P.st[b] = st;
        }
        return;
    }
    // diff_encoder_bb(4): y[n] = (x[n] + y[n-1]) mod 4  ->  exclusive wave scan of the slice sums
    uint32_t incl = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, 64);
        if (lane >= off) incl = (incl + o) & 3u;
    }
    uint32_t run = (st.prev + incl - local) & 3u;                // symbol before this lane's slice
    uint8_t* ring = P.sym.p + (size_t)b * (P.sym.mask + 1u);
    for (uint32_t i = lo; i < hi; ++i) {
        uint32_t reg = 0;
#pragma unroll
        for (int k = 0; k < 7; ++k) reg |= sbit((int64_t)i - k) << k;
        const uint32_t c0 = __builtin_popcount(reg & 109u) & 1u, c1 = __builtin_popcount(reg & 79u) & 1u;
        run = (run + map4[(c0 << 1) | c1]) & 3u;
        ring[(uint32_t)(P.s0 + i) & P.sym.mask] = (uint8_t)run;
    }
    const uint32_t prev_end = __shfl(run, (int)last_lane, 64);
    if (lane == 0 && nbits) {
        uint32_t enc = 0;
        for (int k = 0; k < 6; ++k) enc |= sbit((int64_t)nbits - 1 - k) << k;
        st.sr = sr_end; st.enc = enc; st.prev = prev_end;
        P.st[b] = st;
    }
}

void launch_tx_qpsk_bits(const TxBitsParams& p, int batch, hipStream_t s)
{
    if (!p.nbytes) return;
    const size_t lds = ((size_t)p.nbytes * 8 + 31) / 32 * 4 + 16;
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
