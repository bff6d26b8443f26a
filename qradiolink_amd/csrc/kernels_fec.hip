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

// LANE = TRELLIS STATE, and two trellises share a register: the 8-bit path metrics of unit 2w live in the low and those of unit
// 2w + 1 in the high 16 bits of one VGPR, so one ds_bpermute pair and one v_pk_* add / min serve both.  Unit = (stream, alignment
// branch): with two branches a register pair is branch A and B of one stream, with one branch two neighbouring streams.  The
// arithmetic per trellis is the one stated at the top of the file.  How the kernel got here (profiles/r04_k_fec_rebuild.log):
//   * round 3 ran at the VALU issue rate (71 % of the port) with 21 VALU instructions per step.  Three changes took that to 9:
//     BRANCH METRICS FROM A TABLE -- a state's two addends depend on the step's two soft symbols (wave uniform) and on three bits of
//     the state (the branch-table bits of state >> 1, the state's parity): a pre-pass computes {addend for the lower, for the upper
//     predecessor} of the 8 variants of 8 steps into LDS (7 VALU instructions per 8 steps), a step reads its pair with one
//     ds_read_b64 (8 distinct addresses, a broadcast).  DECISIONS STAY IN THE LANE -- the decision bit of (state, step) is the sign
//     of (lower sum - upper sum); it is shifted into a per-lane history register (no ballot, no SGPR -> VGPR moves, no LDS write,
//     no exec masking).  CHAINBACK ON THE SCALAR UNIT -- the survivor state is wave uniform: a 64-bit scalar shift register whose
//     top six bits are the state takes one v_readlane (history word of lane `state`) and five SALU instructions per step and
//     trellis, and what it shifts out are the decoded bits in order; the descrambler is shifts and xors on those words.
//   * after that four extra VALU or SALU instructions per step cost 1 - 3 %: the step is bound by the LDS pipe (per pair and step
//     two ds_bpermute and one ds_read_b64, 32 waves per CU: ~ 12 LDS cycles x 32 = 384 of the ~ 400 ticks a step takes), with the
//     VALU port (9 - 10 instructions x 8 waves x 4 cycles) close behind -- trading one for the other gains nothing.  The
//     predecessor fetch of step j + 1 is issued BEFORE the renormalisation test of step j (the rare renormalisation hands its
//     per-trellis constant to the next step's adds).
//   * 512 BYTES OF LDS PER WAVE.  In the QPSK receiver this kernel runs beside k_qpsk_pipe4, whose one workgroup per CU holds 137 of
//     the CU's 160 KB for 2 ms.  With the table covering half a block and the block's symbols in LDS (3.8 KB per wave) five waves
//     fit beside it and the decoder starved.  Now the block's 172 symbols stay in three registers (lane t of register r = symbol
//     64 r + t; the pre-pass fetches its two with ds_bpermute, which allocates nothing) and the table covers ONE pass of 8 steps:
//     32 waves = 16 KB.
#ifdef QRL_FEC_PROF
// developer build (tools/kernel_variants.sh kernels_fec.hip name -DQRL_FEC_PROF): shader-clock ticks per phase, summed over every wave
__device__ unsigned long long g_fec_prof[4096][8];
#define FEC_STAMP(k) do { const unsigned long long tn_ = __builtin_readcyclecounter(); pc[k] += tn_ - tprev; tprev = tn_; } while (0)
#else
#define FEC_STAMP(k) do { } while (0)
#endif
__global__ __launch_bounds__(64, 8) void k_fec(const FecParams P, int nunits)
{
    __shared__ uint2 T[64];                      // [step in pass][variant] {addend of the lower predecessor's metric, of the upper one's}
    const int lane = threadIdx.x;
    const int i = lane >> 1, odd = lane & 1;
    // variant of this state: bit 2 = branch-table bit of polynomial 109, bit 1 = of 79 (state >> 1), bit 0 = parity of the state
    const int var = ((__builtin_popcount((2 * i) & 109) & 1) << 2) | ((__builtin_popcount((2 * i) & 79) & 1) << 1) | odd;
    // pre-pass: lane l fills entry l = (step l >> 3 of the pass, variant l & 7)
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
#ifdef QRL_FEC_PROF
    unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_readcyclecounter();
#endif
    for (;;) {
        const bool go0 = valid[0] && st[0].consumed + 172 <= avail[0];
        const bool go1 = valid[1] && st[1].consumed + 172 <= avail[1];
        if (!go0 && !go1) break;
        // the block's soft symbols: lane t of sreg[r] = symbol 64 r + t (unit 1 << 16 | unit 0), zero behind the 172nd
        uint32_t sreg[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int t = lane + 64 * r;
            const int64_t v0 = (int64_t)(st[0].consumed + t) - ubr[0], v1 = (int64_t)(st[1].consumed + t) - ubr[1];
            const uint32_t s0 = (go0 && t < 172 && v0 >= 0) ? soft[0][(uint32_t)v0 & P.soft.mask] : 0u;
            const uint32_t s1 = (go1 && t < 172 && v1 >= 0) ? soft[1][(uint32_t)v1 & P.soft.mask] : 0u;
            sreg[r] = s0 | (s1 << 16);
        }
        uint32_t X = ((lane == (int)(st[0].start_state & 63u)) ? 0u : 63u) | (((lane == (int)(st[1].start_state & 63u)) ? 0u : 63u) << 16);
        FEC_STAMP(0);
        // decision histories, one register per trellis and third of the block (steps 0..31, 32..63, 64..85): bit (last step of the
        // third - step) = "the lower predecessor won"
        uint32_t h0 = 0, h1 = 0, hA0 = 0, hA1 = 0, hB0 = 0, hB1 = 0;
        uint32_t nsub = 0;                       // minus what the renormalisation of the step before subtracted (per-trellis constants, wave uniform)
        uint32_t xi = (uint32_t)__shfl((int)X, i, 64), xj = (uint32_t)__shfl((int)X, i + 32, 64);
#pragma unroll 1
        for (int third = 0; third < 3; ++third) {
            const uint32_t sr = third == 0 ? sreg[0] : third == 1 ? sreg[1] : sreg[2];   // steps 32 third .. + 31 = symbols 64 third .. + 63
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                if (third == 2 && pass == 3) break;                                       // steps 64..85: two passes of 8 and one of 6
                __syncthreads();
                {
                    const int sl = 16 * pass + 2 * (lane >> 3);                           // lane of the pass' step (lane >> 3), first symbol
                    const uint32_t sx = (uint32_t)__shfl((int)sr, sl, 64), sy = (uint32_t)__shfl((int)sr, sl + 1, 64);
                    const uint32_t a = pbt0 ^ sx, c = pbt1 ^ sy;
                    const uint32_t metric = ((a + c + 0x00010001u) >> 3) & 0x003f003fu;   // per half ((a + c + 1) >> 1) >> 2, & 63
                    const uint32_t lo = metric ^ podd;                                     // odd states: 63 - metric from the lower predecessor
                    T[lane] = make_uint2(lo, lo ^ 0x003f003fu);
                }
                __syncthreads();
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int j = 8 * pass + jj;                                          // step 32 third + j
                    if (third == 2 && j == 22) break;
                    const uint2 add = T[8 * jj + var];
                    // VOLK: ma = sat255(xi + a), mb = sat255(xj + b), survivor = min(ma, mb), the upper predecessor wins unless ma < mb.
                    // With ua = xi + a left unsaturated: min(ua, mb) = min(ma, mb), and ua < mb <=> ma < mb (mb <= 255) -- one v_pk_min less.
                    // xi / xj were fetched from the metrics BEFORE the renormalisation of the step before: + nsub puts that right (the
                    // halves never borrow: a trellis' minimum is subtracted from its own metrics).
                    const uint32_t ua = xi + add.x + nsub;
                    const uint32_t mb = pk_min_u16(xj + add.y + nsub, 0x00ff00ffu);
                    X = pk_min_u16(mb, ua);
                    xi = (uint32_t)__shfl((int)X, i, 64); xj = (uint32_t)__shfl((int)X, i + 32, 64);   // next step's predecessors, ahead of the test below
                    const uint32_t z = pk_sub_u16(ua, mb);                 // sign of a half: the LOWER predecessor wins (ties go to the upper one)
                    // (the empty asm pins the update to its step: left alone, the compiler sinks them behind the loop and keeps every ua / mb alive)
                    h1 = __builtin_amdgcn_alignbit(h1, z, 31); h0 = __builtin_amdgcn_alignbit(h0, z << 16, 31); asm volatile("" : "+v"(h1), "+v"(h0));
                    const uint32_t x0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)X);
                    nsub = 0u;
                    if ((x0 + 0x7f2d7f2du) & 0x80008000u) {   // metric[0] > 210 in one of the trellises: renormalise that one (subtract its minimum).
                        // The metrics themselves are only read again through the fetch above (already issued; the next step adds nsub
                        // instead) and by the end-state search, hence the subtraction from X as well.
                        const uint32_t mn = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_min_pk_u16(X));
                        const uint32_t sub = ((x0 & 0xffffu) > 210u ? mn & 0xffffu : 0u) | ((x0 >> 16) > 210u ? mn & 0xffff0000u : 0u);
                        nsub = 0u - sub;
                        X -= sub;
                    }
                }
            }
            if (third == 0) { hA0 = h0; hA1 = h1; } else if (third == 1) { hB0 = h0; hB1 = h1; }
            FEC_STAMP(2);
        }
        // histories hold "lower predecessor wins"; the chainback wants the decision bit (upper wins)
        hA0 = ~hA0; hA1 = ~hA1; hB0 = ~hB0; hB1 = ~hB1; h0 = ~h0; h1 = ~h1;
        asm volatile("" : "+v"(hA0), "+v"(hA1), "+v"(hB0), "+v"(hB1), "+v"(h0), "+v"(h1));   // (6 v_not here, not 160 s_not behind the v_readlanes)
        const uint32_t end0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_min_u32_uniform(((X & 0xffffu) << 6) | (uint32_t)lane)) & 63u;
        const uint32_t end1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_min_u32_uniform(((X >> 16) << 6) | (uint32_t)lane)) & 63u;
        // chainback on the scalar unit: a 64-bit shift register per trellis whose top six bits are the survivor state (wave uniform);
        // after the step of bit nb, bit 63 - j = decoded bit nb + j.  (64 bits wide on purpose: the 32-bit form of this update is a funnel
        // shift, which the compiler can only select as v_alignbit -- VALU -- and then pays a v_readfirstlane per step for the lane index)
        uint64_t SV0 = (uint64_t)end0 << 58, SV1 = (uint64_t)end1 << 58;
        FEC_STAMP(3);
        uint32_t A0 = 0, A1 = 0, next0 = 0, next1 = 0;
#pragma unroll
        for (int nb = 79; nb >= 0; --nb) {
            const int s = nb + 6, th = s >> 5, j = s & 31;
            const int pos = th < 2 ? 31 - j : 21 - j;
            const uint32_t r0 = th == 0 ? hA0 : th == 1 ? hB0 : h0;
            const uint32_t r1 = th == 0 ? hA1 : th == 1 ? hB1 : h1;
            const uint32_t k0 = ((uint32_t)__builtin_amdgcn_readlane((int)r0, (int)(uint32_t)(SV0 >> 58)) >> pos) & 1u;
            const uint32_t k1 = ((uint32_t)__builtin_amdgcn_readlane((int)r1, (int)(uint32_t)(SV1 >> 58)) >> pos) & 1u;
            SV0 = (SV0 >> 1) | ((uint64_t)k0 << 63);
            SV1 = (SV1 >> 1) | ((uint64_t)k1 << 63);
            if (nb == 74) { next0 = (uint32_t)(SV0 >> 58); next1 = (uint32_t)(SV1 >> 58); }
            if (nb == 48) { A0 = (uint32_t)(SV0 >> 32); A1 = (uint32_t)(SV1 >> 32); }   // bit 31 - j = decoded bit 48 + j
        }
        FEC_STAMP(4);
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
    // (a build with two register pairs -- four trellises -- per wave, their chains interleaved, was measured on C5's receiver: 2 264
    // against 1 801 us.  The same LDS-pipe work with half the waves to hide it: profiles/r04_k_fec_rebuild.log.)
    hipLaunchKernelGGL(k_fec, dim3((nunits + 1) / 2), dim3(64), 0, s, p, nunits);
}

}  // namespace qrl
