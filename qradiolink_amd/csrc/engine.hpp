// engine.hpp — device-side parameter blocks and kernel launch entry points of the RX engine.
// Every inter-kernel stream is a per-stream ring in HBM addressed by ABSOLUTE item index
// (idx & mask), so FIR history, loop state and block alignment carry across process() calls
// and results do not depend on how the caller cuts the stream into chunks.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

namespace qrl {

// Raises a kernel's dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize).  The attribute belongs to the CURRENT
// DEVICE: it is set once per (kernel, device), from any thread; the HIP error is returned, never swallowed.
hipError_t dyn_lds_limit(const void* kernel, int bytes);
// launch_* helpers that could not set the attribute skip their launch and leave a per-thread mark; every process() entry point
// ends with take_launch_error() and turns it into QRL_ERR_HIP.
bool take_launch_error();

// A handle-internal stream, optionally confined to a set of compute units (round 6, VERDICT r5 #4: partition the chip in SPACE).
// QRL_CU_<ROLE>="first:count" (CUs PER XCD, 32 each on MI355X) turns the stream of that role into a CU-masked one
// (hipExtStreamCreateWithCUMask); unset = hipStreamCreateWithPriority as before.  Mask bit i = CU (i / 8) of XCD (i % 8): the driver
// deals the bits of a queue's mask round-robin over the XCDs, so a per-XCD range [first, first + count) is bits 8 * first ... 8 * (first + count) - 1.
// A masked stream has a hardware queue of its own (the mask is a queue property), so it cannot share one with another stream of the handle.
int create_role_stream(hipStream_t* s, int priority, const char* role);

struct RingC { float2* p; uint32_t mask; };   // complex ring, stream stride = mask+1 items
struct RingF { float* p; uint32_t mask; };
struct RingB { uint8_t* p; uint32_t mask; };

// ---- K1: rotator + decimating FIR (rotator_cc + rational_resampler_ccf(1,D)) ----
struct DecimParams {
    const float2* in; size_t in_stride;  // caller IQ (or nullptr when in_ring is used)
    RingC in_ring;                       // alternative input: an engine ring (second-stage decimators)
    uint64_t n0; uint32_t n;             // absolute index of in[0], samples in this call
    const float2* hist; uint32_t hist_len;  // last hist_len (rotated) samples before n0, per stream
    RingC out;
    uint64_t m0; uint32_t m_count;       // absolute first output, number of outputs
    const float* taps;                   // [D][Jpad], taps[p*Jpad + j] = h[p + j*D], zero padded
    int D, Jpad;
    int rot_enable; uint64_t rot_acc; uint64_t rot_inc; uint64_t rot_nbase; const float2* rot_lo;
    uint32_t tiles;                      // tiles per stream
    // MFMA variant (kernels_decim_mfma.hip): banded-Toeplitz A operands [S steps][64 lanes], S steps of 4
    const float* gtab; int S; int nt; uint32_t magic_blk, magic_seg;   // gtab: zero-padded taps, hp[k + 4S - nt + 1] = h[k]
    uint32_t tpw, nchunks; int nhi; int dbg;               // consecutive tiles per workgroup; rotator coarse-table entries
    int nld;                                                // 16-byte loads per thread a tile needs (<= the kernel's NLD)
    int hist_raw;                                           // hist holds UN-rotated samples (MFMA variant only): rotate on fetch
    uint32_t out_row_mul_m1, out_row_add;                   // output ring row of stream b = b * (mul_m1 + 1) + add (MFMA / phase-lane variants)
    // phase-lane variant (kernels_decim_pl.hip): lane tap table [J][64]; the launcher fills the segment geometry
    const float* pl_taps; int pl_J; uint32_t pl_S, pl_nseg, pl_batch; uint64_t pl_m_begin, pl_m_end;
    int pl_E, pl_R; const float* pl_hraw;                   // samples per lane and block, outputs per block; raw taps h[k] (k_decim_pl_gen)
    // edge segment (outputs whose window starts in front of this call's buffer): per-stream scratch of ROTATED samples (carried
    // history + head of the buffer), one extra unit per stream behind the regular ones reads it with identity phasors
    float2* pl_edge; uint32_t pl_edge_stride, pl_edge_cap; uint64_t pl_edge_ms, pl_edge_me;
    // host side only: when set, the edge scratch is staged on THIS stream (k_pl_edge_stage reads the caller's buffer and the carried history,
    // nothing the call before produces on the launch stream), `pre_event` is recorded behind it and the launch stream waits for that
    hipStream_t pre_stream; hipEvent_t pre_event;
};
struct HistParams {
    const float2* in; size_t in_stride; uint64_t n0; uint32_t n;
    const float2* hist_old; float2* hist_new; uint32_t hist_len;
    int rot_enable; uint64_t rot_acc; uint64_t rot_inc; uint64_t rot_nbase; const float2* rot_lo;
};
void launch_decim(const DecimParams& p, int batch, int variant, hipStream_t s);
// 1:2 decimator + the channel FIR behind it in one kernel (kernels_frontend.hip k_dec2_fir): d = the decimator's input side (out unused)
struct Dec2FirParams { DecimParams d; const float* taps; RingC out; float2* port; size_t port_cap; uint32_t* counts; };
bool dec2_fir_supported(int nt1, int D, int nt2);
uint32_t dec2_fir_lookback();
std::vector<float> dec2_fir_table(const std::vector<float>& h1, const std::vector<float>& h2);
void launch_dec2_fir(const Dec2FirParams& p, int batch, hipStream_t s);
void launch_hist_save(const HistParams& p, int batch, hipStream_t s);
size_t decim_lds_bytes(int D, int Jpad, int variant);
enum { DECIM_R4_J44 = 0, DECIM_R4_J12 = 1, DECIM_R2_J10 = 2, DECIM_R1_J14 = 3 };
int decim_jc(int variant);
// f32-MFMA decimator (D >= 8): contract "m16" of oracle/orc_blocks.c
bool decim_uses_mfma(int nt, int D);
int decim_mfma_steps(int nt, int D);
int decim_mfma_na(int nt, int D);
int decim_mfma_hpn(int nt, int D);
size_t decim_mfma_lds_bytes(int nt, int D);
int launch_decim_mfma(const DecimParams& p, int batch, hipStream_t s);   // 0, or -1 when the kernel attribute could not be set
void decim_mfma_prof_read(unsigned long long* out8);
void decim_mfma_prof_enable(int on);
// register-resident phase-lane decimator (32 < D <= 64, <= 16 taps per phase): contract "pl" of oracle/orc_blocks.c
bool decim_uses_pl(int nt, int D);
size_t decim_pl_edge_len(int nt, int D);   // samples of edge scratch per stream (0: geometry without the register kernel)
std::vector<float> decim_pl_layout(const std::vector<float>& h, int D);
int launch_decim_pl(const DecimParams& p, int batch, hipStream_t s);
// phase-major matrix-pipe decimator (32 < D <= 52, <= 16 taps per phase): contract "pm" of oracle/orc_blocks.c
bool decim_uses_pm(int nt, int D);
size_t decim_pm_edge_len(int nt, int D);
uint32_t decim_pm_lookback(int nt, int D);
std::vector<float> decim_pm_layout(const std::vector<float>& h, int D);
int launch_decim_pm(const DecimParams& p, int batch, hipStream_t s);

// ---- K2: rational resampler I/D on a ring (optionally with rotator on a caller buffer) ----
struct ResampParams {
    const float2* in; size_t in_stride; uint64_t n0; uint32_t n;  // caller IQ path (rotator-only front end)
    const float2* hist; uint32_t hist_len;
    RingC in_ring;                        // ring path
    RingC out;
    uint64_t q0; uint32_t q_count;        // outputs to produce
    const float* taps;                    // [I][Jp]: taps[ph*Jp + j] = h[ph + j*I]
    int I, D, Jp;
    int rot_enable; uint64_t rot_acc; uint64_t rot_inc; uint64_t rot_nbase; const float2* rot_lo;
    float2* port; size_t port_cap; uint32_t* port_counts;   // optional copy of this call's outputs to a caller port (counts[b*4+0])
};
void launch_resamp(const ResampParams& p, int batch, hipStream_t s);

// ---- small feed-forward kernels at the decimated rate ----
struct FirCcfParams { RingC in; RingC out; uint64_t q0; uint32_t count; const float* taps; int nt;
                      float2* port; size_t port_cap; uint32_t* counts; };  // optional copy to a caller port buffer; counts[b*4+0]
struct FirFffParams { RingF in; RingF out; uint64_t q0; uint32_t count; const float* taps; int nt; };
struct QuadDemodParams { RingC in; RingF out; uint64_t q0; uint32_t count; float gain; const float* atan_tab;
                         RingF out2; float gain2;      // out2.p != nullptr: a second discriminator with its own gain on the same input (C4: MMDVM FM path + 4FSK tail)
                         int16_t* s16; size_t s16_cap; float s16_level, s16_scale; uint32_t* s16_counts; };   // s16 != nullptr: + multiply_const_ff(level) + float_to_short(scale) of `out` (k_f2s fused)
struct Disc2fskParams { RingC in; RingF out; uint64_t q0; uint32_t count; const float2* up; const float2* lo; int nt; };
struct Disc4fskParams { RingC in; RingC out; uint64_t q0; uint32_t count; const float2* taps; int nt; };   // taps[4][nt]
void launch_disc_4fsk(const Disc4fskParams& p, int batch, hipStream_t s);
struct Fsk2FfParams { RingC in; RingF out; uint64_t q0; uint32_t count;
                      const float* tf; int nf; const float2* up; const float2* lo; int nb; const float* ts; int ns;
                      float2* port; size_t port_cap; uint32_t* counts; };
void launch_2fsk_ff(const Fsk2FfParams& p, int batch, hipStream_t s);   // nf/nb/ns and the tables zero padded: fsk2_ff_padded()
int fsk2_ff_padded(int n);
void launch_fir_ccf(const FirCcfParams& p, int batch, hipStream_t s);
void launch_fir_fff(const FirFffParams& p, int batch, hipStream_t s);
void launch_quad_demod(const QuadDemodParams& p, int batch, hipStream_t s);
void launch_disc_2fsk(const Disc2fskParams& p, int batch, hipStream_t s);

// ---- serial loops, one lane per stream ----
struct FllState { float phase, freq; float2 dl[32]; };
struct FllParams { RingC in; RingC out; uint64_t q0; uint32_t count; FllState* st;
                   const float2* lower; const float2* upper; int nt; float alpha, beta, max_freq;
                   int slim; };   // 1: single-wave workgroups with a 16-sample window (3 KB LDS): fits beside a CU full of front-end workgroups (overlapped mode)
void launch_fll(const FllParams& p, int batch, hipStream_t s);

struct SymSyncState { uint64_t ii; uint64_t oo; float mu, avg, inst; float x0, x1, x2, d0, d1, d2; };
struct SymSyncParams {
    RingF in; uint64_t avail;              // samples available (absolute count)
    RingB soft;                            // soft symbols out (absolute symbol index)
    SymSyncState* st;
    const float* mmse;                     // 129 x 8
    float alpha, beta, maxp, minp;
    int ted; float soft_mul, soft_add;
    int slicer;                            // 0: bpsk sign, 1: constellation_rect{-1.5,-0.5,0.5,1.5}
    float tail_scale;                      // tail 1: _level_control in front of the phase modulator (0.9 gr_demod_dmr, 1 gr_demod_m17)
    int tail;                              // 0: soft symbols for the Viterbi; 1: DMR / M17 tail (x scale, phase_modulator, slicer, map) -> bits port;
                                           // 2: native 4FSK (FM) tail: phase_modulator -> (imag, real) soft pairs for the Viterbi
    uint8_t* bits; size_t bits_cap;        // tail 1: two bits per symbol, counts[b*4+2]
    float2* port; size_t port_cap; uint32_t* counts;  // constellation port (this call), counts[b*4+1]
    int slim;                              // 1: the <16 streams, 96-sample window> geometry, 25 KB of LDS (multi-carrier receiver)
};
void launch_symsync_ff(const SymSyncParams& p, int batch, hipStream_t s);

// ---- QPSK recursive chain (agc2 -> costas -> symbol_sync_cc -> costas -> diff_phasor -> rotate), one lane per stream ----
struct QpskState {
    uint64_t ii, oo;                 // symbol-sync read cursor (absolute sample), symbols produced
    float gain;                      // agc2
    float c1_phase, c1_freq;         // first Costas loop (sample rate)
    float mu, avg, inst;             // clock tracking loop
    float2 x0, x1, x2, d0, d1, d2;   // TED history
    float c2_phase, c2_freq;         // second Costas loop (symbol rate)
    float2 dprev;                    // diff_phasor
    float2 hist[16];                 // last 16 outputs of the first Costas loop
};
struct QpskParams {
    RingC in; uint64_t np0, avail;   // RRC-filtered input ring; samples processed before this call / available now
    RingB soft; QpskState* st;
    const float* mmse; const float* tanh_tab;
    float c1_alpha, c1_beta, c2_alpha, c2_beta;
    float ss_alpha, ss_beta, ss_maxp, ss_minp;
    float2 rot; float soft_mul, soft_add;
    int mode;                        // 0: gr_demod_qpsk chain; 1: gr_demod_bpsk chain (agc2 -> clock_recovery_mm_cc -> costas order 2);
                                     // 2: symbol_sync_cc alone on the 4-level rect constellation (gr_demod_4fsk non-FM branch)
    float cr_gain_omega, cr_gain_mu, cr_omega_mid, cr_omega_lim;   // mode 1
    float2* port; size_t port_cap; uint32_t* counts;   // constellation port (this call), counts[b*4+1]
    uint64_t* oo_snap;               // [batch] symbols produced up to the end of this call (what this call's decoder may consume)
};
void launch_qpsk_loops(const QpskParams& p, int batch, hipStream_t s);

// ---- FEC tail: K=7 r=1/2 Viterbi (spiral-kernel semantics) + descrambler ----
struct FecState { uint64_t consumed; uint32_t start_state; uint32_t last_bits; };
struct FecParams {
    RingB soft; const uint64_t* avail; size_t avail_stride; uint32_t avail_mul;   // available soft symbols of stream b = *(avail + b*stride bytes) * mul
    FecState* st;                          // [batch][2]
    uint8_t* bits_a; uint8_t* bits_b; size_t bits_cap; uint32_t* counts;  // counts[b*4+2], [b*4+3]
    int branches;
};
void launch_fec(const FecParams& p, int batch, hipStream_t s);
void launch_fec_gate(unsigned us, hipStream_t s);

// ---- gr_dmr_dmo_sink on the device (kernels_dmo.hip): correlator slicer behind port 3 of gr_demod_dmr ----
struct DmoState {
    uint32_t bitBuffer[5];
    uint16_t syncPtr, startPtr, endPtr, pad0;
    float maxCorr, centre[4], threshold[4];
    uint8_t averagePtr, syncCount, state, control, n, colorCode, pad1[2];
};
struct DmoParams { RingF in; uint64_t q0; uint32_t count; DmoState* st; const uint32_t* golay; uint8_t* out; uint32_t cap; uint32_t* counts; };
void launch_dmo_sink(const DmoParams& p, int batch, hipStream_t s);
std::vector<uint32_t> golay1987_table();

// ---- gr_deframer_bb on the device (kernels_deframe.hip) ----
struct DeframeState { uint32_t reg, found, idx, pad; };
struct DeframeParams {
    const uint8_t* bits; size_t stride; uint32_t n;          // unpacked bits [batch][stride]; n valid per stream unless counts != NULL
    const uint32_t* counts; size_t count_stride;             // device-side valid counts: counts[b * count_stride]
    int type; uint32_t buf_len; DeframeState* st;
    uint8_t* out; size_t out_cap; uint32_t* out_counts;
};
void launch_deframe(const DeframeParams& p, int batch, hipStream_t s);
struct FrameSyncState { uint32_t reg, found, idx, ftype, modem_sync, pad[3]; };
struct FrameSyncParams {
    const uint8_t* bits; size_t stride; uint32_t n; const uint32_t* counts; size_t count_stride;
    int cls; uint32_t bit_buf_len, frame_length;           // sync-word class (0: 1k modes, 1: fast modes, 2: rest), mode table
    FrameSyncState* st; uint8_t* bitbuf; size_t bitbuf_stride;
    uint8_t* out; size_t out_cap; uint32_t* out_counts;    // records; out_counts[2b] = bytes, [2b + 1] = frames
    uint32_t* activity;                                    // optional [batch]: bits collected into a frame while a sync was held, this call
};
void launch_framesync(const FrameSyncParams& p, int batch, hipStream_t s);

// ---- multi-carrier MMDVM RX (kernels_chan.hip) ----
struct ChanParams {
    const float2* in; size_t in_stride; uint64_t n0; uint32_t n;   // wideband caller IQ of this call (n multiple of M)
    const float2* hist; uint32_t hist_len;                         // last hist_len samples before n0
    RingC out; uint64_t m0; uint32_t m_count;                      // channel rings [batch * c_count], output instants
    const float* taps; const float2* twiddle;                      // taps[p + M k] zero padded to J*M; W[q] = e^{+j 2 pi q / M}
    int M, J, c_first, c_count;
    int legacy;                 // qrl_chan_set_option(QRL_CHAN_OPT_LEGACY_PFB), tests: 1 = the general-M kernel also for M = 64
    // row_cpd > 0: output rows grouped by destination rank -- row = ((cc / row_cpd) * batch + b) * row_cpd + cc % row_cpd
    // (the send layout of an all-to-all that gives rank r the channels [r row_cpd, (r + 1) row_cpd) of every stream);
    // out_pitch > 0: linear rows of out_pitch items, item m - m0 (a caller buffer instead of an engine ring)
    uint32_t row_cpd; size_t out_pitch;
};
struct F2sParams { RingF in; uint64_t q0; uint32_t count; float level, scale; int16_t* out; size_t cap; uint32_t* counts; };
struct RssiParams { RingC in; uint64_t j0; uint32_t count; float calibration; float* out; size_t cap; uint32_t* counts; };
void launch_rssi_tag(const RssiParams& p, int batch, hipStream_t s);
// the per-channel chain of gr_demod_mmdvm_multi2 behind the channelizer as one kernel (kernels_chan_tail.hip): 24/25 resampler,
// channel filter, RSSI tags, FM discriminator -> int16, and the symbol demodulator's discriminator + RRC into the symbol-sync ring
struct ChanTailParams {
    RingC in;                               // channel ring at 25 ksps, one row per (stream, channel)
    // form 3 (round 5): the call's channel samples where the exchange put them, lin[row * lin_pitch + (a - lin_base)] for absolute items
    // a in [lin_base, lin_base + lin_n), and the hist_len items in front of them in hist[row * hist_len + ...] (k_hist keeps them from call to
    // call) -- no copy into a ring.  lin == nullptr: the ring.
    const float2* lin; size_t lin_pitch; uint64_t lin_base; uint32_t lin_n; const float2* hist; uint32_t hist_len;
    uint64_t q0; uint32_t count;            // outputs of this call at 24 ksps: [q0, q0 + count)
    const float* tab_a; const float* tab_b; const float* tab_e;   // step-major tap tables (chan_tail_tables; tab_e unused when out_sym.p == nullptr)
    const float* atan_tab;                  // 257
    float gain, gain2, level, scale;
    int16_t* s16; size_t s16_cap; uint32_t* s16_counts;
    RingF out_sym;                          // RRC output (symbol sync input); p == nullptr: no symbol tail
    float* rssi; size_t rssi_cap; uint32_t* rssi_counts; float rssi_cal; uint64_t tag0; uint32_t ntags;
};
void launch_chan_tail(const ChanTailParams& p, int streams, hipStream_t s);
bool chan_tail_supported(int rs_I, int rs_D, int rs_Jp, int filt_nt, int rrc_nt);
std::vector<float> chan_tail_tables(int which, const float* taps);   // 0: resampler (phase-major taps [24][35]), 1: channel filter, 2: RRC
uint32_t chan_tail_lookback();
void launch_pfb_chan(const ChanParams& p, int batch, hipStream_t s);
void launch_ring_store(RingC in, uint64_t q0, uint32_t count, float2* out, size_t cap, uint32_t* counts, int rows, hipStream_t s);
void launch_ring_load(const float2* in, size_t pitch, RingC out, uint64_t q0, uint32_t count, int rows, hipStream_t s);
void launch_f2s(const F2sParams& p, int batch, hipStream_t s);
size_t chan_lds_bytes(int M, int J);
// ---- multi-carrier MMDVM TX (kernels_chan.hip) ----
struct S2fInParams { const int16_t* in; size_t in_stride; RingF out; uint64_t q0; uint32_t count; float scale, level; };
struct SynthParams {
    RingC in; int nch;                                   // channel rings [batch * nch] at 25 ksps
    int port_chan[16];                                   // port p <- channel ring port_chan[p], -1 = idle port
    uint64_t blk0; uint32_t nblk;                        // absolute first block, blocks of this call
    const float* taps; const float2* twiddle; int M, J;  // taps[i + M j] zero padded to J*M; W[q] = e^{+j 2 pi q / M}
    float level, bb_gain;
    float2* out; size_t out_stride, out_cap;
};
void launch_s2f_in(const S2fInParams& p, int streams, hipStream_t s);
void launch_scale_c(RingC r, uint64_t q0, uint32_t count, float k, int streams, hipStream_t s);
struct ZeroRun { uint32_t row; uint32_t pad; uint64_t start, count; };   // ring row, absolute item range [start, start + count)
void launch_zero_runs(RingC r, const ZeroRun* runs, uint32_t nruns, uint64_t lo, uint64_t hi, hipStream_t s);   // gr_zero_idle_bursts
void launch_pfb_synth(const SynthParams& p, int batch, hipStream_t s);
size_t synth_lds_bytes(int M, int J);

// ---- TX: gr_mod_qpsk (kernels_tx.hip) ----
struct TxState { uint32_t sr, enc, prev, pad; };   // scrambler register, last 6 scrambled bits, last differential symbol
struct TxBitsParams {
    const uint8_t* bytes; size_t stride; uint32_t nbytes; uint32_t L;   // L bits per lane (multiple of 32)
    uint8_t tl_pow[6][8];                                                // T^(L 2^d), d = 0..5, of the zero-input scrambler: column masks
    int mode;                                                            // 0: QPSK differential symbols, 1: coded bits (2 per input bit)
    TxState* st; RingB sym; uint64_t s0;                                 // symbol ring, absolute index of this call's first symbol
};
struct TxInterpParams {
    RingB sym; uint64_t n0; uint32_t count;      // absolute first output sample, outputs of this call
    const float* taps; int nt; int interp; float2 table[4]; float amp, bb_gain;
    float2* out; size_t out_stride;
};
struct TxShapeParams { RingB sym; RingF out; uint64_t n0; uint32_t count; int sps; const float* taps; int nt;   // nt = 0: repeat
                       int levels; float scale; };   // levels 2 | 4; scale 0 = none
struct TxFmParams { RingF in; RingC out; uint64_t n0; uint32_t count; float k, amp; float* phase; };
struct TxInterpCParams { RingC in; uint64_t n0; uint32_t count; const float* taps; int nt; int interp; float2* out; size_t out_stride;
                         int decim; RingC out_ring; };   // out_ring.p != nullptr: the samples go to ring item n0 + t instead of out (gr_mod_am: a filter follows)   // decim > 1: rational_resampler_ccf(interp, decim) (gr_mod_m17: 125 / 3)
void launch_tx_spread(RingB coded, RingB chips, uint64_t c0, uint32_t ncoded, int batch, hipStream_t s);   // gr_mod_dsss: Barker-13 spreading
void launch_tx_f2c(RingF in, RingC out, uint64_t n0, uint32_t count, float g, int batch, hipStream_t s);
void launch_tx_raw_dibits(const uint8_t* bytes, size_t stride, uint32_t nbytes, RingB sym, uint64_t s0, int batch, hipStream_t s);
struct TxRotParams { const float2* in; size_t in_stride; uint64_t n0; uint32_t count; uint64_t rot_acc, rot_inc, rot_nbase; const float2* rot_lo;
                     RingC out_ring; float2* out; size_t out_stride; };
void launch_tx_rot(const TxRotParams& p, int batch, hipStream_t s);
void launch_tx_shape(const TxShapeParams& p, int batch, hipStream_t s);
void launch_tx_fm(const TxFmParams& p, int batch, hipStream_t s);
void launch_tx_interp_c(const TxInterpCParams& p, int batch, hipStream_t s);
void launch_tx_qpsk_bits(const TxBitsParams& p, int batch, hipStream_t s);
void launch_tx_interp(const TxInterpParams& p, int batch, hipStream_t s);

// ---- DSSS mode (kernels_dsss.hip) ----
struct DsssState { float phase, freq, gain, pad; };
struct DsssLoopParams { RingC in, out; uint64_t q0; uint32_t count; DsssState* st; const float* tanh_tab; float alpha, beta; };
void launch_dsss_loop(const DsssLoopParams& p, int mode, int batch, hipStream_t s);   // 0: costas(order 2, snr), 1: agc2(0.1, 0.1)
struct DsssMfParams { RingC in, out; uint64_t i0; uint32_t count; const float* taps; };
void launch_dsss_mf(const DsssMfParams& p, int batch, hipStream_t s);
struct DsssTailState { uint64_t ii, oo; float mu, omega, phase, freq; float2 p0, p1, p2, c0, c1, c2; };
struct DsssTailParams { RingC in; uint64_t avail; RingB soft; DsssTailState* st; const float* mmse;
                        float gain_omega, gain_mu, omega_mid, omega_lim, alpha, beta;
                        float2* port; size_t port_cap; uint32_t* counts; };
void launch_dsss_tail(const DsssTailParams& p, int batch, hipStream_t s);

// ---- analogue voice receivers (kernels_analog.hip) ----
constexpr int AN_MAX_RAMP = 1024;
struct FirCccParams { RingC in, out; uint64_t q0; uint32_t count; const float2* taps; int nt; float2* port; size_t port_cap; uint32_t* counts; };
void launch_an_fir_ccc(const FirCccParams& p, int batch, hipStream_t s);
// g / g_prev: items that passed the gating squelch up to the end of this / the previous call
struct AnState { double pwr, iir_y, de_y; uint64_t g, g_prev; float2 prev; float iir_x, de_x, gain, env; int state, ramped; };
struct AnGateParams { RingC in; RingF out; RingC outc; uint64_t q0; uint32_t count; AnState* st; const float* atan_tab; const float* env; int ramp;
                      double alpha, one_minus_alpha, threshold; float gain, attack, decay, ref, clip; double ff0, ff1, fb1; };
void launch_an_gate(const AnGateParams& p, int kind, int batch, hipStream_t s);   // kind 0 NBFM, 1 AM, 2 WBFM, 3 SSB (complex out)
struct AnResampParams { RingF in, out; const AnState* st; const float* taps; int nt, I, D; float* port; size_t port_cap; uint32_t* counts;
                        uint64_t q0; uint32_t count; };   // st == nullptr: outputs q0 .. q0 + count (host-side counts: the TX chains)
void launch_an_resamp(const AnResampParams& p, uint32_t max_out, int batch, hipStream_t s);
// analog::ctcss_squelch_ff between the audio resampler and the audio filter (gr_demod_nbfm::set_ctcss, gr_demod_nbfm.cpp:59-60,97-123):
// per stream the three Goertzel filters, the squelch_base state machine and the count of items that passed the gate (g2)
struct CtcssState { float d1[3], d2[3]; int processed, mute, state, ramped; double env; uint64_t g2, g2_prev; };
struct CtcssParams { RingF in, out; const AnState* st; CtcssState* cs; int I, D; float wr[3], wi[3]; double level; int len, ramp; const double* env; };
void launch_an_ctcss(const CtcssParams& p, int batch, hipStream_t s);
struct AnFirParams { RingF in, out; const AnState* st; const float* taps; int nt, I, D; float* port; size_t port_cap; uint32_t* counts;
                     const CtcssState* cs; };   // cs != nullptr: the call's output range is the CTCSS gate's (g2_prev .. g2)
void launch_an_fir(const AnFirParams& p, uint32_t max_out, int batch, hipStream_t s);
struct AnDeemphParams { RingF in; AnState* st; int I, D; double ff0, ff1, fb1; float* port; size_t port_cap; uint32_t* counts; const CtcssState* cs; };
void launch_an_deemph(const AnDeemphParams& p, int batch, hipStream_t s);
// I == 0 in AnFirParams / AnStretchParams: the output range of the call is the cessb stretcher's, 1024 floor((g - 2) / 1024)
// analogue modulators (gr_mod_nbfm): linear audio -> ring; x gain -> two-tap IIR in double (lane per stream)
struct AmAgcParams { RingF in, out; uint64_t n0; uint32_t count; float attack, decay, ref, max_gain, lo, hi, scale; float* gain; };   // gr_mod_am: agc2_ff -> rail_ff -> multiply_const_ff
void launch_am_agc_rail(const AmAgcParams& p, int batch, hipStream_t s);
void launch_am_carrier(RingF in, RingC out, uint64_t n0, uint32_t count, float carrier, int batch, hipStream_t s);
struct AmLoadParams { const float* in; size_t in_stride; RingF out; uint64_t n0; uint32_t count; };
void launch_am_load(const AmLoadParams& p, int batch, hipStream_t s);
struct AmToneParams { RingF out; uint64_t n0; uint32_t count; const float* tab; uint32_t inc; uint64_t k0; double ampl; float offset; };
void launch_am_tone(const AmToneParams& p, int batch, hipStream_t s);
struct AmIirState { double y1; float x1, pad; };
struct AmIirParams { RingF in, out; uint64_t n0; uint32_t count; float gain; double ff0, ff1, fb1; AmIirState* st;
                     // gr_mod_nbfm::set_ctcss: add_ff(audio, sig_source_f(8000, GR_COS_WAVE, tone, 0.15)) in front of the filter -- the fixed-point NCO of
                     // oracle/orc_chains.c orc_sig_source_cos: sample k (tone_k0 + t) = (float)(cos_fx(k * tone_inc) * tone_ampl); tone_tab == nullptr: no tone
                     const float* tone_tab; uint32_t tone_inc; uint64_t tone_k0; double tone_ampl; };
void launch_am_iir(const AmIirParams& p, int batch, hipStream_t s);
// gr_mod_ssb: float_to_complex + cessb clipper (f32 ring -> complex ring), cessb stretcher with complex output over [q0, q0 + count)
struct AmClipParams { RingF in; RingC out; uint64_t n0; uint32_t count; float clip; const float* atan_tab; };
void launch_am_clip(const AmClipParams& p, int batch, hipStream_t s);
struct AmStretchParams { RingC in, out; uint64_t q0; uint32_t count; };
void launch_am_stretch(const AmStretchParams& p, int batch, hipStream_t s);
struct AnStretchParams { RingC in; RingF out; const AnState* st; float level; };
void launch_an_stretch(const AnStretchParams& p, uint32_t max_out, int batch, hipStream_t s);

// ---- frame FEC (kernels_framefec.hip): bursts [n][33] <-> payloads [n][12]; M17 frames [n][48] -> records [n][40] ----
void launch_bptc_decode(const uint8_t* bursts, size_t n, uint8_t* payloads, hipStream_t s);
void launch_bptc_encode(const uint8_t* payloads, size_t n, uint8_t* bursts, hipStream_t s);
void launch_m17_decode(const uint8_t* frames, size_t n, uint8_t* records, hipStream_t s);
void launch_m17_encode(const uint8_t* records, size_t n, uint8_t* frames, hipStream_t s);

// ---- side outputs (kernels_side.hip): rssi_block on port 0, rx_fft_c on the device-rate IQ ----
constexpr uint32_t RSSI_RING = 4096;   // |x|^2 look-back ring per stream (moving_average_ff(2000) reads 1999 items back)
struct RssiState { double prev; uint64_t n; float sum, last; };
struct RssiBlockParams {
    const float2* in; size_t in_stride; uint32_t n;       // port-0 items of this call: in[b * in_stride + i]
    const uint32_t* counts; size_t count_stride;         // per-stream item count (device), or nullptr: n for every stream
    RssiState* st; float* ring; int batch;
    float level, n_log2_10;
    float* out; size_t out_cap; float* last; uint32_t* out_counts;
};
void launch_rssi(const RssiBlockParams& p, hipStream_t s);
void launch_fft_fill(const float2* in, size_t in_stride, uint32_t i0, uint32_t count, const float* win, uint32_t counter, float2* buf, uint32_t N, int batch, hipStream_t s);
void launch_fft_power(const float2* X, float* out, uint32_t N, int batch, hipStream_t s);
void launch_fft_shift(const float* pts, float* out, size_t out_stride, uint32_t N, int batch, hipStream_t s);

}  // namespace qrl
