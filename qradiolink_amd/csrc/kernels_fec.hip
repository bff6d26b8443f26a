// kernels_fec.hip — FEC tail: streaming K=7 r=1/2 Viterbi + self-synchronising descrambler.
//   fec::decoder(cc_decoder(80, 7, 2, {109,79})) + descrambler_bb(0x8A, 0x7F, 7)
//   (gr_demod_2fsk.cpp:120-127,155-164; gr_demod_gmsk.cpp:103-111,122-131; gr_demod_qpsk.cpp:124-126)
// One wave64 per two (stream, alignment branch) units: a LANE is a TRELLIS STATE, the two trellises' 8-bit path metrics share a VGPR
// (16-bit halves).  Metric arithmetic restates VOLK's volk_8u_x4_conv_k7_r2_8u_spiral (the variant the reference requires,
// docs/OPERATION.md:4): avg_epu8 branch metric >> 2, saturating u8 adds, ties pick the upper predecessor, renormalise (subtract the
// minimum) only when metric[0] > 210.  Branch B (port 3) decodes the stream delayed by one soft symbol (blocks::delay(1)).
#include "devmath.hpp"
#include "engine.hpp"
#include <type_traits>

namespace qrl {

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

#ifdef QRL_FEC_PROF
// developer build (tools/kernel_variants.sh kernels_fec.hip name -DQRL_FEC_PROF): shader-clock ticks per phase, summed over every wave
__device__ unsigned long long g_fec_prof[4096][8];
#define FEC_STAMP(k) do { const unsigned long long tn_ = __builtin_readcyclecounter(); pc[k] += tn_ - tprev; tprev = tn_; } while (0)
#else
#define FEC_STAMP(k) do { } while (0)
#endif

// ---- the decoder: ROTATING state -> lane layout (round 6) ---------------------------------------------------------------------------
// The decoded bit sequence b_t is what a state is a window of: the state after step s holds b_s .. b_(s-5), b_s at bit 0.  LANE BIT
// (t mod 6) HOLDS b_t for as long as b_t is in the window: a step replaces the oldest bit b_(s-6) by the newest bit b_s in the SAME lane
// bit q = s mod 6, every other bit stays where it is.  The two predecessors of a butterfly are then the two lanes that differ in bit q
// -- a lane distance of 1, 2, 4, 8, 16, 32 in turn -- and a step gets them without the LDS:
//     q = 0, 1   quad_perm DPP folded into the two adds (v_add_u32_dpp)                                        8 VALU per step
//     q = 2, 3   bank-masked row shifts: own sum everywhere, then the other role's lanes take their partner's   10
//     q = 4, 5   both sums of the own metric, then ONE v_permlane16_swap / v_permlane32_swap                     9
// (the rest of a step: saturate the upper sum, survivor = v_pk_min, sign of lower - upper = "the lower predecessor wins" -> shifted into
// a per-lane history register with v_lshrrev + v_and_or -- both trellises at once --, v_readfirstlane of state 0's metric for the
// renormalisation test, which is a scalar branch that is rarely taken.)  Rounds 3 - 5 kept lane = state and fetched the predecessors
// n >> 1 and (n >> 1) + 32 with two ds_bpermute per step: LDS pipe ~ 384 of the ~ 400 ticks of a step at 32 waves per CU, VALU close
// behind, 1 923 us on C5's receiver alone.  This layout: 1 419 us, VALU port ~ 87 % busy (profiles/r06_k_fec_rebuild.log).
//   * BRANCH METRICS FROM A TABLE (round 4) -- a state's two addends depend on the step's two soft symbols (wave uniform) and three bits
//     of the state (the branch-table bits of the butterfly, which of its two outputs): a pre-pass computes {addend of the lower, of the
//     upper predecessor} for the 8 variants of 8 steps into LDS (7 VALU per 8 steps), a step reads its pair with one ds_read_b64 (8
//     distinct addresses, a broadcast) issued a step ahead.  Which variant a lane is depends on q: six precomputed offsets.
//   * CHAINBACK ON THE SCALAR UNIT -- the survivor's lane index p IS the window b_s .. b_(s-5): the step of decoded bit nb reads the
//     history word of lane p (v_readlane), and replaces bit nb mod 6 of p by the decision (three scalar instructions); after every
//     sixth step p holds six decoded bits in order.  The descrambler is shifts and xors on the assembled words.
//   * 512 BYTES OF LDS PER WAVE, the block's 172 symbols in three registers (lane t of register r = symbol 64 r + t), the next block's
//     fetched while this one is decoded: in the QPSK receiver this kernel runs beside k_qpsk_pipe4, whose one workgroup per CU holds
//     137 of the CU's 160 KB for 2 ms.
// ua = (metric of the pair's lower state) + a.x, ub = (metric of its upper state) + a.y, for the pair of lanes that differ in bit Q
template <int Q> __device__ __forceinline__ void fec_sums(uint32_t X, uint2 a, uint32_t& ua, uint32_t& ub)
{
    if constexpr (Q == 0) {          // the DPP moves fold into the adds (v_add_u32_dpp)
        ua = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)X, 0xA0, 0xf, 0xf, true) + a.x;      // quad_perm [0,0,2,2]
        ub = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)X, 0xF5, 0xf, 0xf, true) + a.y;      // quad_perm [1,1,3,3]
    } else if constexpr (Q == 1) {
        ua = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)X, 0x44, 0xf, 0xf, true) + a.x;      // quad_perm [0,1,0,1]
        ub = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)X, 0xEE, 0xf, 0xf, true) + a.y;      // quad_perm [2,3,2,3]
    } else if constexpr (Q == 2) {   // own sum everywhere, then the lanes of the other role take their partner's (bank-masked row shifts; the
                                     // two plain adds in front are also the two wait states a DPP read of a freshly written VGPR needs)
        asm("v_add_u32 %0, %2, %3\n\tv_add_u32 %1, %2, %4\n\t"
            "v_add_u32_dpp %0, %2, %3 row_shr:4 row_mask:0xf bank_mask:0xa\n\tv_add_u32_dpp %1, %2, %4 row_shl:4 row_mask:0xf bank_mask:0x5"
            : "=&v"(ua), "=&v"(ub) : "v"(X), "v"(a.x), "v"(a.y));
    } else if constexpr (Q == 3) {
        asm("v_add_u32 %0, %2, %3\n\tv_add_u32 %1, %2, %4\n\t"
            "v_add_u32_dpp %0, %2, %3 row_shr:8 row_mask:0xf bank_mask:0xc\n\tv_add_u32_dpp %1, %2, %4 row_shl:8 row_mask:0xf bank_mask:0x3"
            : "=&v"(ua), "=&v"(ub) : "v"(X), "v"(a.x), "v"(a.y));
    } else if constexpr (Q == 4) {   // both sums of the own metric first: a pair's two lanes hold each other's addends swapped (their variants differ in
                                     // bit 0 only), so ONE swap hands every lane its pair's lower sum in ua and upper sum in ub
        const auto r = __builtin_amdgcn_permlane16_swap(X + a.x, X + a.y, false, false);        // odd rows of the first <-> even rows of the second
        ua = r[0]; ub = r[1];
    } else {
        const auto r = __builtin_amdgcn_permlane32_swap(X + a.x, X + a.y, false, false);        // upper half of the first <-> lower half of the second
        ua = r[0]; ub = r[1];
    }
}
template <int S, int N, class F> __device__ __forceinline__ void fec_static_for(F&& f)
{
    if constexpr (S < N) { f(std::integral_constant<int, S>{}); fec_static_for<S + 1, N>(f); }
}
__global__ __launch_bounds__(64, 8) void k_fec(const FecParams P, int nunits)
{
    __shared__ uint2 T[64];                      // [step in pass][variant] {addend of the lower predecessor's metric, of the upper one's}
    const int lane = threadIdx.x;
    // variant of this lane at a step with s mod 6 = q: bit 0 = lane bit q (the new bit: which of the butterfly's two outputs), bits 2 / 1 =
    // branch-table bits of butterfly i, whose bit j = b_(s-1-j) sits in lane bit (q - 1 - j) mod 6
    uint32_t var[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        int i = 0;
#pragma unroll
        for (int j = 0; j < 5; ++j) i |= ((lane >> ((q + 5 - j) % 6)) & 1) << j;
        var[q] = ((__builtin_popcount((2 * i) & 109) & 1) << 2) | ((__builtin_popcount((2 * i) & 79) & 1) << 1) | ((lane >> q) & 1);
    }
    // before step 0 the window is b_(-1) .. b_(-6): state bit j = b_(-1-j) in lane bit 5 - j
    const uint32_t lstart = __builtin_bitreverse32((uint32_t)lane) >> 26;
    // after step 85: state bit j = b_(85-j) in lane bit (85 - j) mod 6 = 1, 0, 5, 4, 3, 2
    const uint32_t lend = ((lane >> 1) & 1) | (((lane >> 0) & 1) << 1) | (((lane >> 5) & 1) << 2) | (((lane >> 4) & 1) << 3) | (((lane >> 3) & 1) << 4) | (((lane >> 2) & 1) << 5);
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
    // a block's soft symbols: lane t of sr[r] = symbol 64 r + t (unit 1 << 16 | unit 0), zero behind the 172nd
    auto load_syms = [&](uint64_t c0, uint64_t c1, uint32_t (&sr)[3]) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int t = lane + 64 * r;
            const int64_t v0 = (int64_t)(c0 + t) - ubr[0], v1 = (int64_t)(c1 + t) - ubr[1];
            // (unconditional loads -- the ring index is always inside the ring -- so that all six are in flight together)
            const uint32_t l0 = soft[0][(uint32_t)v0 & P.soft.mask], l1 = soft[1][(uint32_t)v1 & P.soft.mask];
            const uint32_t s0 = (valid[0] && t < 172 && v0 >= 0) ? l0 : 0u;
            const uint32_t s1 = (valid[1] && t < 172 && v1 >= 0) ? l1 : 0u;
            sr[r] = s0 | (s1 << 16);
        }
    };
    uint32_t sreg[3];
    load_syms(st[0].consumed, st[1].consumed, sreg);
#ifdef QRL_FEC_PROF
    unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_readcyclecounter();
#endif
    for (;;) {
        const bool go0 = valid[0] && st[0].consumed + 172 <= avail[0];
        const bool go1 = valid[1] && st[1].consumed + 172 <= avail[1];
        if (!go0 && !go1) break;
        // the next block's symbols are fetched now and used an iteration later (a trellis that stops here keeps its position)
        uint32_t snext[3];
        load_syms(st[0].consumed + (go0 ? 160u : 0u), st[1].consumed + (go1 ? 160u : 0u), snext);
        uint32_t X = ((lstart == (st[0].start_state & 63u)) ? 0u : 63u) | (((lstart == (st[1].start_state & 63u)) ? 0u : 63u) << 16);
        // decision histories: register s >> 4 holds steps 16 (s >> 4) .. + 15 of both trellises, "the lower predecessor won" of step s in
        // bit (s & 15) of each half once the register is complete (the last one holds six steps: bit 10 + (s & 15))
        uint32_t H[6] = {0, 0, 0, 0, 0, 0};
        FEC_STAMP(0);
        uint2 add = make_uint2(0u, 0u);
        fec_static_for<0, 86>([&](auto S_) {
            constexpr int S = decltype(S_)::value;
            constexpr int jj = S & 7;
            if constexpr (jj == 0) {
                const uint32_t sr = S < 32 ? sreg[0] : S < 64 ? sreg[1] : sreg[2];   // steps 32 r .. + 31 = symbols 64 r .. + 63
                __syncthreads();
                {
                    const int sl = 2 * (S & 31) + 2 * (lane >> 3);                    // lane of the pass' step (lane >> 3), first symbol
                    const uint32_t sx = (uint32_t)__shfl((int)sr, sl, 64), sy = (uint32_t)__shfl((int)sr, sl + 1, 64);
                    const uint32_t a = pbt0 ^ sx, c = pbt1 ^ sy;
                    const uint32_t metric = ((a + c + 0x00010001u) >> 3) & 0x003f003fu;   // per half ((a + c + 1) >> 1) >> 2, & 63
                    const uint32_t lo = metric ^ podd;                                     // odd states: 63 - metric from the lower predecessor
                    T[lane] = make_uint2(lo, lo ^ 0x003f003fu);
                }
                __syncthreads();
                add = T[var[S % 6]];
            }
            // VOLK: ma = sat255(lower + a), mb = sat255(upper + b), survivor = min(ma, mb), the upper predecessor wins unless ma < mb.
            // With ua = lower + a left unsaturated: min(ua, mb) = min(ma, mb), and ua < mb <=> ma < mb (mb <= 255) -- one v_pk_min less.
            uint32_t ua, ub;
            fec_sums<S % 6>(X, add, ua, ub);
            if constexpr (jj < 7 && S < 85) add = T[8 * (jj + 1) + var[(S + 1) % 6]];   // the next step's addends, a step ahead
            const uint32_t mb = pk_min_u16(ub, 0x00ff00ffu);
            X = pk_min_u16(mb, ua);
            const uint32_t z = pk_sub_u16(ua, mb);                                        // sign of a half: the LOWER predecessor wins
            H[S >> 4] = (H[S >> 4] >> 1) | (z & 0x80008000u);
            asm volatile("" : "+v"(H[S >> 4]));
            const uint32_t x0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)X);         // lane 0 = state 0 in every layout
            if ((x0 + 0x7f2d7f2du) & 0x80008000u) {   // metric[0] > 210 in one of the trellises: renormalise that one (subtract its minimum)
                const uint32_t mn = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_min_pk_u16(X));
                X -= ((x0 & 0xffffu) > 210u ? mn & 0xffffu : 0u) | ((x0 >> 16) > 210u ? mn & 0xffff0000u : 0u);
            }
        });
        FEC_STAMP(2);
        // histories hold "lower predecessor wins"; the chainback wants the decision bit (upper wins)
#pragma unroll
        for (int r = 0; r < 6; ++r) { H[r] = ~H[r]; asm volatile("" : "+v"(H[r])); }
        // first minimum in STATE order; the low six bits carry the lane it sits in
        uint32_t p0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_min_u32_uniform(((X & 0xffffu) << 12) | (lend << 6) | (uint32_t)lane)) & 63u;
        uint32_t p1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_min_u32_uniform(((X >> 16) << 12) | (lend << 6) | (uint32_t)lane)) & 63u;
        // chainback on the scalar unit: the lane index p is the window b_s .. b_(s-5) (b_t in bit t mod 6); the step of decoded bit nb (trellis
        // step s = nb + 6) replaces bit nb mod 6 -- b_(nb+6) -- by the decision b_nb.  After a step with nb mod 6 = 0, p = b_nb .. b_(nb+5) in order.
        FEC_STAMP(3);
        uint32_t W0[3] = {0, 0, 0}, W1[3] = {0, 0, 0};   // decoded bits 0..29, 30..59, 60..83 of each trellis, LSB first
        fec_static_for<0, 80>([&](auto N_) {
            constexpr int nb = 79 - decltype(N_)::value;
            constexpr int s = nb + 6, r = s >> 4, pos = r < 5 ? (s & 15) : 10 + (s & 15), q = nb % 6;
            const uint32_t k0 = ((uint32_t)__builtin_amdgcn_readlane((int)H[r], (int)p0) >> pos) & 1u;
            const uint32_t k1 = ((uint32_t)__builtin_amdgcn_readlane((int)H[r], (int)p1) >> (pos + 16)) & 1u;
            p0 = (p0 & ~(1u << q)) | (k0 << q);
            p1 = (p1 & ~(1u << q)) | (k1 << q);
            if constexpr (q == 0) { W0[nb / 30] |= p0 << (nb % 30); W1[nb / 30] |= p1 << (nb % 30); }
        });
        FEC_STAMP(4);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (!(q ? go1 : go0)) continue;
            const uint32_t* W = q ? W1 : W0;
            const uint64_t DL = (uint64_t)W[0] | ((uint64_t)W[1] << 30) | ((uint64_t)W[2] << 60);   // decoded bits 0..63
            const uint32_t DH = (W[2] >> 4) & 0xffffu;                                              // 64..79
            // F: bit k + 8 = decoded bit k, bits 0..7 = the last 8 bits of the block before (last_bits bit t = d[-1 - t])
            const uint32_t prev8 = __builtin_bitreverse32(st[q].last_bits) >> 24;
            const uint64_t FL = (uint64_t)prev8 | (DL << 8);
            const uint32_t FH = (uint32_t)(DL >> 56) | (DH << 8);
            // descrambler_bb(0x8A, 0x7F, 7) as restated so far: o[k] = d[k] ^ d[k-1] ^ d[k-5] ^ d[k-7]
            const uint64_t OL = FL ^ (FL << 1) ^ (FL << 5) ^ (FL << 7);
            const uint32_t OH = FH ^ ((FH << 1) | (uint32_t)(FL >> 63)) ^ ((FH << 5) | (uint32_t)(FL >> 59)) ^ ((FH << 7) | (uint32_t)(FL >> 57));
            const uint64_t EL = (OL >> 8) | ((uint64_t)OH << 56);   // bit k = o[k], k < 64
            const uint32_t EH = OH >> 8;                            // bit k - 64, k = 64..79
            if (out[q]) {
                if (nout[q] + lane < P.bits_cap) out[q][nout[q] + lane] = (uint8_t)((EL >> lane) & 1ull);
                if (lane < 16 && nout[q] + 64 + lane < P.bits_cap) out[q][nout[q] + 64 + lane] = (uint8_t)((EH >> lane) & 1u);
            }
            const uint32_t last8 = __builtin_bitreverse32(DH >> 8) >> 24;   // bit t = d[79 - t]
            st[q].last_bits = last8;
            st[q].start_state = last8 & 63u;                        // the state after step 79: bit j = d[79 - j]
            st[q].consumed += 160;
            nout[q] += 80;
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) sreg[r] = snext[r];
        FEC_STAMP(5);
#ifdef QRL_FEC_PROF
        pc[7] += 1;
#endif
    }
#ifdef QRL_FEC_PROF
    if (lane == 0) {
        unsigned long long* slot = g_fec_prof[(blockIdx.x * 61u) & 4095u];
        for (int k = 0; k < 8; ++k) atomicAdd(&slot[k], pc[k]);
    }
#endif
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (!valid[q]) continue;
            P.st[ub[q] * 2 + ubr[q]] = st[q];
            P.counts[ub[q] * 4 + 2 + ubr[q]] = nout[q] < P.bits_cap ? nout[q] : (uint32_t)P.bits_cap;   // what was WRITTEN: consumers (deframer, frame sync) trust it
        }
    }
}

#ifdef QRL_FEC_PROF
extern "C" void qrl_fec_prof_read(unsigned long long* out8)
{
    (void)hipDeviceSynchronize();
    static unsigned long long host[4096][8];
    (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_fec_prof), sizeof host);
    for (int k = 0; k < 8; ++k) { out8[k] = 0; for (int i = 0; i < 4096; ++i) out8[k] += host[i][k]; }
    static unsigned long long zero[4096][8];
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fec_prof), zero, sizeof zero);
}
#endif
// One wave that does nothing for `us` microseconds (constant 100 MHz counter): put in front of the decoder on its stream in the
// grouped order, so that the recursion kernel released by the same event has its workgroups placed before the decoder's waves take
// every wave slot of the chip (they keep them for the decoder's whole run).
__global__ __launch_bounds__(64) void k_fec_gate(unsigned us)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < 100ull * us) __builtin_amdgcn_s_sleep(32);
}
void launch_fec_gate(unsigned us, hipStream_t s) { hipLaunchKernelGGL(k_fec_gate, dim3(1), dim3(64), 0, s, us); }

void launch_fec(const FecParams& p, int batch, hipStream_t s)
{
    const int nunits = batch * p.branches;
    hipLaunchKernelGGL(k_fec, dim3((nunits + 1) / 2), dim3(64), 0, s, p, nunits);
}

}  // namespace qrl
