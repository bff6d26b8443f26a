// kernels_fec.hip — FEC tail: streaming K=7 r=1/2 Viterbi + self-synchronising descrambler.
//   fec::decoder(cc_decoder(80, 7, 2, {109,79})) + descrambler_bb(0x8A, 0x7F, 7)
//   (gr_demod_2fsk.cpp:120-127,155-164; gr_demod_gmsk.cpp:103-111,122-131; gr_demod_qpsk.cpp:124-126)
// One wave64 per (stream, alignment branch): LANE = TRELLIS STATE.  Path metrics live in one VGPR,
// the two predecessors of state n (n>>1 and (n>>1)+32) arrive by ds_bpermute, the 64 decision bits
// of a trellis step are one __ballot -> one 64-bit word, exactly the decision-word layout the
// chainback walks.  Metric arithmetic restates VOLK's volk_8u_x4_conv_k7_r2_8u_spiral (the variant
// the reference requires, docs/OPERATION.md:4): avg_epu8 branch metric >> 2, saturating u8 adds,
// ties pick the upper predecessor, renormalise (subtract min) only when metric[0] > 210.
// Branch B (port 3) decodes the stream delayed by one soft symbol (blocks::delay(1)).
#include "devmath.hpp"
#include "engine.hpp"

namespace qrl {

__device__ __forceinline__ int wave_min_i32(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}

__global__ __launch_bounds__(64) void k_fec(const FecParams P)
{
    __shared__ unsigned long long dec[86];
    __shared__ uint8_t sym[176];
    __shared__ uint8_t dbits[96];
    const int b = blockIdx.x, br = blockIdx.y, lane = threadIdx.x;
    FecState st = P.st[b * 2 + br];
    const uint64_t avail = *reinterpret_cast<const uint64_t*>(reinterpret_cast<const char*>(P.avail) + (size_t)b * P.avail_stride) * P.avail_mul + (uint64_t)br;
    const uint8_t* soft = P.soft.p + (size_t)b * (P.soft.mask + 1u);
    uint8_t* out = (br ? P.bits_b : P.bits_a);
    if (out) out += (size_t)b * P.bits_cap;
    const int i = lane >> 1, odd = lane & 1;
    const int bt0 = (__builtin_popcount((2 * i) & 109) & 1) ? 255 : 0;
    const int bt1 = (__builtin_popcount((2 * i) & 79) & 1) ? 255 : 0;
    uint32_t nout = 0;
    while (st.consumed + 172 <= avail) {
        __syncthreads();
        for (int t = lane; t < 172; t += 64) {
            const int64_t v = (int64_t)(st.consumed + t) - br;
            sym[t] = (v >= 0) ? soft[(uint32_t)v & P.soft.mask] : (uint8_t)0;
        }
        if (lane < 8) dbits[7 - lane] = (st.last_bits >> lane) & 1u;  // dbits[8 - t] = d[-t], t = lane + 1
        __syncthreads();
        int X = (lane == (int)(st.start_state & 63u)) ? 0 : 63;
        for (int s = 0; s < 86; ++s) {
            const int a = bt0 ^ (int)sym[2 * s];
            const int c = bt1 ^ (int)sym[2 * s + 1];
            const int metric = (((a + c + 1) >> 1) >> 2) & 63;
            const int xi = __shfl(X, i, 64);
            const int xj = __shfl(X, i + 32, 64);
            int ma = xi + (odd ? 63 - metric : metric);
            int mb = xj + (odd ? metric : 63 - metric);
            ma = ma > 255 ? 255 : ma;
            mb = mb > 255 ? 255 : mb;
            const int surv = mb < ma ? mb : ma;
            const unsigned long long bal = __ballot(surv == mb);
            if (lane == 0) dec[s] = bal;
            X = surv;
            if (__builtin_amdgcn_readfirstlane(X) > 210) X -= wave_min_i32(X);
        }
        const int end = wave_min_i32((X << 6) | lane) & 63;
        __syncthreads();
        int next = 0;
        if (lane == 0) {
            int sv = end;
            for (int nb = 79; nb >= 0; --nb) {
                const int k = (int)((dec[nb + 6] >> sv) & 1ull);
                sv = (sv >> 1) | (k << 5);
                dbits[8 + nb] = (uint8_t)k;
                if (nb == 74) next = sv;
            }
        }
        next = __builtin_amdgcn_readfirstlane(next);
        __syncthreads();
        for (int k = lane; k < 80; k += 64) {
            const uint8_t o = dbits[8 + k] ^ dbits[8 + k - 1] ^ dbits[8 + k - 5] ^ dbits[8 + k - 7];
            if (out && nout + k < P.bits_cap) out[nout + k] = o;
        }
        uint32_t lb = 0;
#pragma unroll
        for (int t = 0; t < 8; ++t) lb |= (uint32_t)dbits[8 + 79 - t] << t;
        st.last_bits = lb;
        st.start_state = (uint32_t)next;
        st.consumed += 160;
        nout += 80;
    }
    if (lane == 0) {
        P.st[b * 2 + br] = st;
        P.counts[b * 4 + 2 + br] = nout < P.bits_cap ? nout : (uint32_t)P.bits_cap;   // what was WRITTEN: consumers (deframer, frame sync) trust it
    }
}

void launch_fec(const FecParams& p, int batch, hipStream_t s)
{
    hipLaunchKernelGGL(k_fec, dim3(batch, p.branches), dim3(64), 0, s, p);
}

}  // namespace qrl
