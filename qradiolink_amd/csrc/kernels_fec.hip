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

// Minimum over the wave of a u32, result uniform: two quad butterflies, row_half_mirror and row_mirror leave every row of 16 lanes
// holding its minimum (DPP folded into v_min_u32), then one lane of each row is read back and the four are reduced on the SALU.
__device__ __forceinline__ uint32_t wave_min_u32_uniform(uint32_t v)
{
    uint32_t o;
    o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xf, 0xf, false); v = o < v ? o : v;    // quad_perm [1,0,3,2]
    o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xf, 0xf, false); v = o < v ? o : v;    // quad_perm [2,3,0,1]
    o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xf, 0xf, false); v = o < v ? o : v;   // row_half_mirror
    o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xf, 0xf, false); v = o < v ? o : v;   // row_mirror
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    const uint32_t ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}
__device__ __forceinline__ uint32_t pk_sub_u16(uint32_t a, uint32_t b)
{
    const us2 r = __builtin_bit_cast(us2, a) - __builtin_bit_cast(us2, b);
    return __builtin_bit_cast(uint32_t, r);
}

// TWO decoders per wave, LANE = TRELLIS STATE: the 8-bit path metrics of unit 2w live in the low and those of unit 2w + 1 in the high
// 16 bits of one VGPR, so one ds_bpermute pair and one v_pk_* add / min serve both trellises.  Unit = (stream, alignment branch):
// with two branches the pair is branch A and B of one stream, with one branch two neighbouring streams.  The arithmetic per trellis
// is the one stated at the top of the file; what this kernel is organised around is the VALU instruction count per trellis step
// (the kernel runs at the VALU issue rate: round-3 counters, 71 % of the port with 21 VALU instructions per step):
//   * BRANCH METRICS FROM A TABLE.  A state's two metric addends depend on the step's two soft symbols (wave uniform) and on three
//     bits of the state (the two branch-table bits of state >> 1 and the state's parity): 8 variants per step and trellis pair.  A
//     pre-pass computes {add for the lower, add for the upper predecessor} of the 8 variants of 48 steps (6 x 7 VALU instructions for
//     43 steps) into LDS; a step reads its pair with one ds_read_b64 (8 distinct addresses: a broadcast) instead of computing it
//     with 8 VALU instructions.  The table covers half a block (43 steps), so that 8 waves per SIMD still fit the LDS.
//   * DECISIONS STAY IN THE LANE.  The decision bit of (state, step) is the sign of (lower sum - upper sum) after both have been
//     saturated; it is shifted into a per-lane history register (v_alignbit / v_bfe + v_lshl_or: 4 instructions for the two
//     trellises) -- no ballot, no SGPR -> VGPR moves, no LDS write, no exec masking.
//   * CHAINBACK ON THE SCALAR UNIT.  The survivor state is wave uniform: a 32-bit scalar shift register whose top six bits are the
//     state takes one v_readlane (history word of lane `state`) and five SALU instructions per step and trellis, and what it shifts
//     out are the decoded bits in order -- three snapshots hold the block's 80 bits; the descrambler is shifts and xors on them.
constexpr int FEC_HALF = 43;      // trellis steps per table pass (86 per block)
constexpr int FEC_TSTEPS = 48;    // steps the pre-pass covers (6 iterations of 8 steps)
__global__ __launch_bounds__(64, 8) void k_fec(const FecParams P, int nunits)
{
    __shared__ uint2 symp2[96];                  // soft symbol pair of step s: .x = symbol 2 s, .y = symbol 2 s + 1 (trellis 1 << 16 | trellis 0); 86 used, the pre-pass reads up to 91
    __shared__ uint2 T[FEC_TSTEPS * 8];          // [step in half][variant] {addend of the lower predecessor's metric, of the upper one's}
    uint32_t* symp = reinterpret_cast<uint32_t*>(symp2);
    const int lane = threadIdx.x;
    const int i = lane >> 1, odd = lane & 1;
    // variant of this state: bit 2 = branch-table bit of polynomial 109, bit 1 = of 79 (state >> 1), bit 0 = parity of the state
    const int var = ((__builtin_popcount((2 * i) & 109) & 1) << 2) | ((__builtin_popcount((2 * i) & 79) & 1) << 1) | odd;
    const uint2* Tl = T + var;
    // pre-pass: lane l fills entry l + 64 j = (step (l >> 3) + 8 j, variant l & 7)
    const uint32_t pbt0 = (lane & 4) ? 0x00ff00ffu : 0u, pbt1 = (lane & 2) ? 0x00ff00ffu : 0u, podd = (lane & 1) ? 0x003f003fu : 0u;
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
    if (lane < 20) symp[172 + lane] = 0u;        // read by the pre-pass of the second half beyond the block's 172 symbols, never used
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
        uint32_t X = ((lane == (int)(st[0].start_state & 63u)) ? 0u : 63u) | (((lane == (int)(st[1].start_state & 63u)) ? 0u : 63u) << 16);
        // decision histories: [trellis][a = steps 0..31 of the half, b = steps 32..42]; g* = first half of the block, h* = second
        uint32_t h0a = 0, h0b = 0, h1a = 0, h1b = 0, g0a = 0, g0b = 0, g1a = 0, g1b = 0;
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            __syncthreads();
            {
                const uint2* sp = symp2 + half * FEC_HALF + (lane >> 3);
#pragma unroll
                for (int j = 0; j < FEC_TSTEPS / 8; ++j) {
                    const uint2 sy = sp[8 * j];
                    const uint32_t a = pbt0 ^ sy.x, c = pbt1 ^ sy.y;
                    const uint32_t metric = ((a + c + 0x00010001u) >> 3) & 0x003f003fu;   // per half ((a + c + 1) >> 1) >> 2, & 63
                    const uint32_t lo = metric ^ podd;                                     // odd states: 63 - metric from the lower predecessor
                    T[lane + 64 * j] = make_uint2(lo, lo ^ 0x003f003fu);
                }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < FEC_HALF; ++j) {
                const uint2 add = Tl[8 * j];
#if defined(QRL_FEC_ABL) && QRL_FEC_ABL == 1   // developer ablation (wrong results): the two predecessor fetches as DPP moves instead of ds_bpermute
                const uint32_t xi = (uint32_t)__builtin_amdgcn_update_dpp((int)X, (int)X, 0x121, 0xf, 0xf, false);
                const uint32_t xj = (uint32_t)__builtin_amdgcn_update_dpp((int)X, (int)X, 0x122, 0xf, 0xf, false);
#else
                const uint32_t xi = (uint32_t)__shfl((int)X, i, 64);
                const uint32_t xj = (uint32_t)__shfl((int)X, i + 32, 64);
#endif
                const uint32_t ma = pk_min_u16(xi + add.x, 0x00ff00ffu);   // saturating u8 adds
                const uint32_t mb = pk_min_u16(xj + add.y, 0x00ff00ffu);
                X = pk_min_u16(mb, ma);
                const uint32_t z = pk_sub_u16(ma, mb);                     // sign of a half: the LOWER predecessor wins (ties go to the upper one)
                // (the empty asm pins the update to its step: left alone, the compiler sinks all 43 behind the loop and keeps every ma / mb alive)
                if (j < 32) { h1a = __builtin_amdgcn_alignbit(h1a, z, 31); h0a = __builtin_amdgcn_alignbit(h0a, z << 16, 31); asm volatile("" : "+v"(h1a), "+v"(h0a)); }
                else        { h1b = __builtin_amdgcn_alignbit(h1b, z, 31); h0b = __builtin_amdgcn_alignbit(h0b, z << 16, 31); asm volatile("" : "+v"(h1b), "+v"(h0b)); }
                const uint32_t x0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)X);
                if ((x0 + 0x7f2d7f2du) & 0x80008000u) {   // metric[0] > 210 in one of the trellises: renormalise that one (subtract its minimum)
                    const uint32_t mn = wave_min_pk_u16(X);
                    X -= ((x0 & 0xffffu) > 210u ? mn & 0xffffu : 0u) | ((x0 >> 16) > 210u ? mn & 0xffff0000u : 0u);
                }
            }
            if (half == 0) { g0a = h0a; g0b = h0b; g1a = h1a; g1b = h1b; }
        }
        // histories hold "lower predecessor wins"; the chainback wants the decision bit (upper wins)
        g0a = ~g0a; g0b = ~g0b; g1a = ~g1a; g1b = ~g1b; h0a = ~h0a; h0b = ~h0b; h1a = ~h1a; h1b = ~h1b;
        asm volatile("" : "+v"(g0a), "+v"(g0b), "+v"(g1a), "+v"(g1b), "+v"(h0a), "+v"(h0b), "+v"(h1a), "+v"(h1b));   // (8 v_not here, not 160 s_not behind the v_readlanes)
        const uint32_t end0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_min_u32_uniform(((X & 0xffffu) << 6) | (uint32_t)lane)) & 63u;
        const uint32_t end1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_min_u32_uniform(((X >> 16) << 6) | (uint32_t)lane)) & 63u;
        // chainback on the scalar unit: a 64-bit shift register per trellis whose top six bits are the survivor state (wave uniform);
        // after the step of bit nb, bit 63 - j = decoded bit nb + j.  (64 bits wide on purpose: the 32-bit form of this update is a funnel
        // shift, which the compiler can only select as v_alignbit -- VALU -- and then pays a v_readfirstlane per step for the lane index)
        uint64_t SV0 = (uint64_t)end0 << 58, SV1 = (uint64_t)end1 << 58;
        uint32_t A0 = 0, A1 = 0, next0 = 0, next1 = 0;
#pragma unroll
        for (int nb = 79; nb >= 0; --nb) {
            const int s = nb + 6, hf = s >= FEC_HALF ? 1 : 0, j = s - FEC_HALF * hf;
            const int pos = j < 32 ? 31 - j : FEC_HALF - 1 - j;
            const uint32_t r0 = hf ? (j < 32 ? h0a : h0b) : (j < 32 ? g0a : g0b);
            const uint32_t r1 = hf ? (j < 32 ? h1a : h1b) : (j < 32 ? g1a : g1b);
            const uint32_t k0 = ((uint32_t)__builtin_amdgcn_readlane((int)r0, (int)(uint32_t)(SV0 >> 58)) >> pos) & 1u;
            const uint32_t k1 = ((uint32_t)__builtin_amdgcn_readlane((int)r1, (int)(uint32_t)(SV1 >> 58)) >> pos) & 1u;
            SV0 = (SV0 >> 1) | ((uint64_t)k0 << 63);
            SV1 = (SV1 >> 1) | ((uint64_t)k1 << 63);
            if (nb == 74) { next0 = (uint32_t)(SV0 >> 58); next1 = (uint32_t)(SV1 >> 58); }
            if (nb == 48) { A0 = (uint32_t)(SV0 >> 32); A1 = (uint32_t)(SV1 >> 32); }   // bit 31 - j = decoded bit 48 + j
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (!(q ? go1 : go0)) continue;
            const uint64_t SV = q ? SV1 : SV0;                       // bit 63 - j = decoded bit j
            const uint32_t SA = q ? A1 : A0;
            const uint32_t D0 = __builtin_bitreverse32((uint32_t)(SV >> 32)), D1 = __builtin_bitreverse32((uint32_t)SV);   // decoded bits 0..31, 32..63
            const uint32_t D2 = __builtin_bitreverse32(SA) >> 16;                                                           // 64..79
            // F: bit k + 8 = decoded bit k, bits 0..7 = the last 8 bits of the block before (last_bits bit t = d[-1 - t])
            const uint32_t prev8 = __builtin_bitreverse32(st[q].last_bits) >> 24;
            const uint64_t FL = (uint64_t)prev8 | ((uint64_t)D0 << 8) | ((uint64_t)D1 << 40);
            const uint32_t FH = (D1 >> 24) | (D2 << 8);
            // descrambler_bb(0x8A, 0x7F, 7) as restated so far: o[k] = d[k] ^ d[k-1] ^ d[k-5] ^ d[k-7]
            const uint64_t OL = FL ^ (FL << 1) ^ (FL << 5) ^ (FL << 7);
            const uint32_t OH = FH ^ ((FH << 1) | (uint32_t)(FL >> 63)) ^ ((FH << 5) | (uint32_t)(FL >> 59)) ^ ((FH << 7) | (uint32_t)(FL >> 57));
            const uint64_t EL = (OL >> 8) | ((uint64_t)OH << 56);   // bit k = o[k], k < 64
            const uint32_t EH = OH >> 8;                            // bit k - 64, k = 64..79
            if (out[q]) {
                if (nout[q] + lane < P.bits_cap) out[q][nout[q] + lane] = (uint8_t)((EL >> lane) & 1ull);
                if (lane < 16 && nout[q] + 64 + lane < P.bits_cap) out[q][nout[q] + 64 + lane] = (uint8_t)((EH >> lane) & 1u);
            }
            st[q].last_bits = SA & 0xffu;                           // bit t = d[79 - t]
            st[q].start_state = q ? next1 : next0;
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
