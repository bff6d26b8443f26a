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

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b)
{
    const us2 r = __builtin_elementwise_min(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b));
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t wave_min_pk_u16(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = pk_min_u16(v, (uint32_t)__shfl_xor((int)v, off, 64));
    return v;
}

// TWO decoders per wave: the 8-bit path metrics of unit 2w live in the low and those of unit 2w + 1 in the high 16 bits of one
// VGPR, so one ds_bpermute pair and one v_pk_* add / min serve both trellises (a wave with one 8-bit metric per lane spends the
// same ~30 instructions per trellis step on a quarter of the register).  Unit = (stream, alignment branch): with two branches
// the pair is branch A and B of one stream, with one branch two neighbouring streams.  The arithmetic per trellis is unchanged.
__global__ __launch_bounds__(64) void k_fec(const FecParams P, int nunits)
{
    __shared__ unsigned long long dec[2][86];
    __shared__ uint32_t symp[176];
    __shared__ uint8_t dbits[2][96];
    const int lane = threadIdx.x;
    const int i = lane >> 1, odd = lane & 1;
    const uint32_t bt0 = (__builtin_popcount((2 * i) & 109) & 1) ? 0x00ff00ffu : 0u;
    const uint32_t bt1 = (__builtin_popcount((2 * i) & 79) & 1) ? 0x00ff00ffu : 0u;
    FecState st[2];
    uint64_t avail[2];
    const uint8_t* soft[2];
    uint8_t* out[2];
    int ub[2], ubr[2];
    bool valid[2];
    uint32_t nout[2] = {0, 0};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int u = blockIdx.x * 2 + q;
        valid[q] = u < nunits;
        const int uu = valid[q] ? u : 0;
        ub[q] = P.branches == 2 ? uu >> 1 : uu;
        ubr[q] = P.branches == 2 ? uu & 1 : 0;
        st[q] = P.st[ub[q] * 2 + ubr[q]];
        avail[q] = *reinterpret_cast<const uint64_t*>(reinterpret_cast<const char*>(P.avail) + (size_t)ub[q] * P.avail_stride) * P.avail_mul + (uint64_t)ubr[q];
        soft[q] = P.soft.p + (size_t)ub[q] * (P.soft.mask + 1u);
        out[q] = ubr[q] ? P.bits_b : P.bits_a;
        if (out[q]) out[q] += (size_t)ub[q] * P.bits_cap;
    }
    for (;;) {
        const bool go0 = valid[0] && st[0].consumed + 172 <= avail[0];
        const bool go1 = valid[1] && st[1].consumed + 172 <= avail[1];
        if (!go0 && !go1) break;
        __syncthreads();
        for (int t = lane; t < 172; t += 64) {
            const int64_t v0 = (int64_t)(st[0].consumed + t) - ubr[0], v1 = (int64_t)(st[1].consumed + t) - ubr[1];
            const uint32_t s0 = (go0 && v0 >= 0) ? soft[0][(uint32_t)v0 & P.soft.mask] : 0u;
            const uint32_t s1 = (go1 && v1 >= 0) ? soft[1][(uint32_t)v1 & P.soft.mask] : 0u;
            symp[t] = s0 | (s1 << 16);
        }
        if (lane < 8) {   // dbits[8 - t] = d[-t], t = lane + 1
            dbits[0][7 - lane] = (st[0].last_bits >> lane) & 1u;
            dbits[1][7 - lane] = (st[1].last_bits >> lane) & 1u;
        }
        __syncthreads();
        uint32_t X = ((lane == (int)(st[0].start_state & 63u)) ? 0u : 63u) | (((lane == (int)(st[1].start_state & 63u)) ? 0u : 63u) << 16);
        for (int s = 0; s < 86; ++s) {
            const uint32_t a = bt0 ^ symp[2 * s];
            const uint32_t c = bt1 ^ symp[2 * s + 1];
            const uint32_t metric = ((a + c + 0x00010001u) >> 3) & 0x003f003fu;   // per half ((a + c + 1) >> 1) >> 2, & 63
            const uint32_t minv = 0x003f003fu - metric;
            const uint32_t xi = (uint32_t)__shfl((int)X, i, 64);
            const uint32_t xj = (uint32_t)__shfl((int)X, i + 32, 64);
            uint32_t ma = xi + (odd ? minv : metric);
            uint32_t mb = xj + (odd ? metric : minv);
            ma = pk_min_u16(ma, 0x00ff00ffu);   // saturating u8 adds
            mb = pk_min_u16(mb, 0x00ff00ffu);
            const uint32_t surv = pk_min_u16(mb, ma);
            const uint32_t d = surv ^ mb;       // half == 0: the upper predecessor wins (ties too)
            const unsigned long long bal0 = __ballot((d & 0xffffu) == 0u);
            const unsigned long long bal1 = __ballot((d >> 16) == 0u);
            if (lane == 0) { dec[0][s] = bal0; dec[1][s] = bal1; }
            X = surv;
            const uint32_t x0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)X);
            if ((x0 & 0xffffu) > 210u || (x0 >> 16) > 210u) {   // renormalise (subtract the minimum) the trellis whose metric[0] > 210
                const uint32_t mn = wave_min_pk_u16(X);
                X -= ((x0 & 0xffffu) > 210u ? mn & 0xffffu : 0u) | ((x0 >> 16) > 210u ? mn & 0xffff0000u : 0u);
            }
        }
        const int end0 = wave_min_i32((int)((X & 0xffffu) << 6) | lane) & 63;
        const int end1 = wave_min_i32((int)((X >> 16) << 6) | lane) & 63;
        __syncthreads();
        int next = 0;
        if (lane < 2) {   // chainback of both trellises side by side on lanes 0 and 1
            int sv = lane ? end1 : end0;
            for (int nb = 79; nb >= 0; --nb) {
                const int k = (int)((dec[lane][nb + 6] >> sv) & 1ull);
                sv = (sv >> 1) | (k << 5);
                dbits[lane][8 + nb] = (uint8_t)k;
                if (nb == 74) next = sv;
            }
        }
        const int next0 = __builtin_amdgcn_readlane(next, 0), next1 = __builtin_amdgcn_readlane(next, 1);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (!(q ? go1 : go0)) continue;
            for (int k = lane; k < 80; k += 64) {
                const uint8_t o = dbits[q][8 + k] ^ dbits[q][8 + k - 1] ^ dbits[q][8 + k - 5] ^ dbits[q][8 + k - 7];
                if (out[q] && nout[q] + k < P.bits_cap) out[q][nout[q] + k] = o;
            }
            uint32_t lb = 0;
#pragma unroll
            for (int t = 0; t < 8; ++t) lb |= (uint32_t)dbits[q][8 + 79 - t] << t;
            st[q].last_bits = lb;
            st[q].start_state = (uint32_t)(q ? next1 : next0);
            st[q].consumed += 160;
            nout[q] += 80;
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (!valid[q]) continue;
            P.st[ub[q] * 2 + ubr[q]] = st[q];
            P.counts[ub[q] * 4 + 2 + ubr[q]] = nout[q] < P.bits_cap ? nout[q] : (uint32_t)P.bits_cap;   // what was WRITTEN: consumers (deframer, frame sync) trust it
        }
    }
}

void launch_fec(const FecParams& p, int batch, hipStream_t s)
{
    const int nunits = batch * p.branches;
    hipLaunchKernelGGL(k_fec, dim3((nunits + 1) / 2), dim3(64), 0, s, p, nunits);
}

}  // namespace qrl
