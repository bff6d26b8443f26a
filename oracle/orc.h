/*
 * orc.h — CPU ORACLE for the gr_modem RX/TX DSP hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product library (libqrl_hip.so)
 * never links, loads or calls anything in this directory.
 *
 * PARITY UNPINNED: the reference (qradiolink @ 2025-06-14) ships no tests, golden vectors
 * or fixtures, and its arithmetic lives in un-vendored GNU Radio 3.10 / VOLK, which is
 * not installable here.  This oracle restates the published
 * GNU Radio 3.10 block semantics from the upstream algorithm descriptions; every chain
 * function cites the reference file:line whose topology and parameters it follows.
 * It is pinned only by (i) constants quoted from upstream (tap counts, MMSE rows, atan /
 * tanh table endpoints) and (ii) mod -> channel -> demod loopback known-answer tests.
 * EXCEPTION -- pinned against the real reference: what the reference implements in its OWN plain C++ is compiled from where it
 * lies (make -C oracle ref -> oracle/_ref/libqrl_ref.so, built from /root/reference sources + oracle/gr_stub) and the
 * restatement must reproduce it exactly: gr_dmr_dmo_sink (orc_dmr.c), gr_deframer_bb, gr_4fsk_discriminator, rssi_tag_block,
 * gr_zero_idle_bursts, dsss_decoder_cc / dsss_encoder_bb, cessb clipper / stretcher, calculate_deemph_taps, gr_modem (orc_modem_sync; host/gr_modem_hip.cpp),
 * BurstTimer + gr_mmdvm_sink + gr_mmdvm_source (host/mmdvm_wire.cpp), CBPTC19696 + CHamming, M17FrameDecoder + M17Viterbi + Golay(24,12) (orc_framefec.c):
 * tests/test_ref_blocks.py, tests/test_ref_mmdvm.py, tests/test_ref_modem.py, tests/test_framefec.py, tests/golden/ref/.
 * ALSO pinned: the CONSTRUCTION of every chain (which stock blocks, every parameter, the firdes call behind every tap vector, the
 * wiring) -- the reference's gr_demod_*.cpp / gr_mod_*.cpp constructors run unmodified against recording stand-ins of the GNU Radio
 * factories (oracle/rec_stub, oracle/_ref/libqrl_rec.so) and must list what the chains here trace (orc_trace.c):
 * tests/test_ref_chains.py.  What the stock blocks COMPUTE stays unpinned.
 *
 * ARITHMETIC CONTRACT (what makes GPU results bit-identical to this oracle):
 *   - IEEE-754 binary32, round-to-nearest-even, no flush-to-zero, compiled with
 *     -ffp-contract=off: a fused multiply-add happens only where fmaf() is written.
 *   - FIR dot products are fmaf chains in the order documented at each function
 *     (VOLK leaves the summation order to the SIMD ISA; we fix one).
 *   - run-time sin/cos (NCOs of rotator, FLL, Costas) use orc_sincosf()/orc_sincos_turn(),
 *     a fixed polynomial, so results do not depend on the libm of the machine.
 *   - design-time constants (taps, loop gains, tables) are computed in double and
 *     rounded once to float.
 */
#ifndef ORC_H
#define ORC_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
/* ---- construction trace (orc_trace.c): which primitives a chain calls, with which parameters; tests/test_ref_chains.py ---- */
void orc_trace_enable(int on);            /* clears the trace; on != 0 starts recording */
int  orc_trace_on(void);
const char* orc_trace_get(void);          /* one "name(args)" line per primitive call */
void orc_trace_event(const char* fmt, ...);
void orc_trace_taps(const void* taps, size_t bytes, const char* fmt, ...);
const char* orc_trace_name(const void* taps, size_t bytes);
/* block timing for bench.py's cpu_baseline "thread_per_block_model" (orc_trace.c): run time of every primitive call of one chain run */
void orc_block_timing_enable(int on);
int  orc_block_timing_count(void);
double orc_block_timing_get(int i, char* name, size_t cap);

#endif

typedef struct { float re, im; } cf32;

/* ---------- deterministic math ---------- */
void  orc_sincosf(float x, float* s, float* c);            /* |x| <= ~8 rad */
void  orc_sincos_turn(uint64_t angle, float* s, float* c); /* angle in 2^-64 turns */
float orc_fast_atan2f(float y, float x);                   /* gnuradio fast_atan2f */
float orc_tanhf_lut(float x);                              /* gnuradio tanhf_lut */
const float* orc_atan_table(void);  /* 257 entries */
const float* orc_tanh_table(void);  /* 256 entries */
const float* orc_mmse_table(void);  /* 129 x 8 */

/* ---------- firdes (gr-filter/lib/firdes.cc, gr-fft/lib/window.cc) ---------- */
enum { ORC_WIN_HAMMING = 0, ORC_WIN_HANN = 1, ORC_WIN_BLACKMAN = 2, ORC_WIN_RECTANGULAR = 3,
       ORC_WIN_BLACKMAN_HARRIS = 5 };
int orc_window(int type, int ntaps, float* w);
int orc_compute_ntaps(double fs, double tw, int win);
int orc_compute_ntaps_windes(double fs, double tw, double atten_db);
/* each returns ntaps; if taps==NULL only the count is returned */
int orc_low_pass(double gain, double fs, double fc, double tw, int win, float* taps);
int orc_low_pass_2(double gain, double fs, double fc, double tw, double atten_db, int win, float* taps);
int orc_complex_band_pass(double gain, double fs, double lo, double hi, double tw, int win, cf32* taps);
int orc_root_raised_cosine(double gain, double fs, double symrate, double alpha, int ntaps, float* taps);
int orc_gaussian(double gain, double spb, double bt, int ntaps, float* taps);
void orc_fll_taps(float sps, float rolloff, int n, cf32* lower, cf32* upper);
void orc_control_loop_gains(float bw, float* alpha, float* beta);
void orc_clock_loop_gains(float loop_bw, float zeta, float ted_gain, float* alpha, float* beta);

/* ---------- stream blocks (one-shot over a finite stream, zero initial state) ---------- */
uint64_t orc_phase_inc_to_turn(double radians_per_sample);
void   orc_rotator(const cf32* in, size_t n, uint64_t inc, uint64_t acc0, cf32* out);
size_t orc_decim_count(size_t n, int interp, int decim);
size_t orc_decim_fir_ccf(const cf32* in, size_t n, const float* taps, int nt, int decim, int nsplit, cf32* out);
size_t orc_decim_fir_ccf_m16(const cf32* in, size_t n, const float* taps, int nt, int decim, cf32* out);
int    orc_m16_steps(int nt, int decim);
int    orc_decim_uses_m16(int nt, int decim);
size_t orc_decim_fir_ccf_pl(const cf32* in, size_t n, const float* taps, int nt, int decim, cf32* out);
void   orc_pl_geometry(int decim, int* dprime, int* per_lane);
int    orc_decim_uses_pl(int nt, int decim);
size_t orc_decim_fir_ccf_pm(const cf32* in, size_t n, const float* taps, int nt, int decim, cf32* out);   /* phase-major matrix-pipe contract, 32 < D <= 64 */
int    orc_decim_uses_pm(int nt, int decim);
size_t orc_decim_fir_ccf_simd(const cf32* in, size_t n, const float* taps, int nt, int decim, cf32* out);   /* CPU baseline only */
void   orc_set_decim_impl(int impl);   /* 0: summation contracts (checker), 1: AVX2 dot product (bench.py cpu_baseline timing) */
size_t orc_decim_xlating(const cf32* in, size_t n, const float* taps, int nt, int D, cf32* out);   /* multi-carrier graphs: always the m16 order */
size_t orc_decim_auto(const cf32* in, size_t n, const float* taps, int nt, int decim, cf32* out);
size_t orc_resamp_ccf(const cf32* in, size_t n, const float* taps, int nt, int interp, int decim, cf32* out);
size_t orc_resamp_fff(const float* in, size_t n, const float* taps, int nt, int interp, int decim, float* out);
void   orc_fir_ccf(const cf32* in, size_t n, const float* taps, int nt, cf32* out);
void   orc_fir_ccc(const cf32* in, size_t n, const cf32* taps, int nt, cf32* out);
void   orc_fir_ccc_conj_pair(const cf32* in, size_t n, const cf32* up, const cf32* lo, int nt, cf32* out_up, cf32* out_lo);   /* conjugate tap pair: shared real-tap chains */
void   orc_fir_fff(const float* in, size_t n, const float* taps, int nt, float* out);
void   orc_fll_band_edge(const cf32* in, size_t n, float sps, float rolloff, int ntaps, float bw, cf32* out);
void   orc_quad_demod(const cf32* in, size_t n, float gain, float* out);
void   orc_agc2(const cf32* in, size_t n, float attack, float decay, float ref, float gain, float max_gain, cf32* out);
void   orc_costas(const cf32* in, size_t n, float bw, int order, int use_snr, cf32* out);
enum { ORC_TED_MM = 0, ORC_TED_MOD_MM = 1 };
/* modified-M&M error formula: a named contract of include/qrl_contracts.h (QRL_TED_MODMM_*); < 0 restores the contract default.
 * Process-wide; for the sensitivity test only (tests/test_ted_sensitivity.py). */
void orc_set_ted_modmm(int ff_variant, int cc_variant);
void orc_get_ted_modmm(int* ff_variant, int* cc_variant);
enum { ORC_CONST_BPSK = 0, ORC_CONST_DQPSK = 1, ORC_CONST_4LEVEL = 2 };
size_t orc_loop_clamp_hits(int reset);   /* test statistic: limiter hits of the timing loops since the last reset */
size_t orc_symbol_sync_ff(const float* in, size_t n, int ted, float sps, float loop_bw, float damping,
                          float ted_gain, float max_dev, int constellation, float* out);
size_t orc_symbol_sync_cc(const cf32* in, size_t n, int ted, float sps, float loop_bw, float damping,
                          float ted_gain, float max_dev, int constellation, cf32* out);
void   orc_diff_phasor(const cf32* in, size_t n, cf32* out);
void   orc_soft_quant(const float* in, size_t n, float mul, float add, uint8_t* out);
size_t orc_cc_decode_k7(const uint8_t* soft, size_t n, uint8_t* bits);   /* streaming, frame 80 */
void   orc_descramble(const uint8_t* in, size_t n, uint32_t mask, uint32_t seed, int len, uint8_t* out);
void   orc_scramble(const uint8_t* in, size_t n, uint32_t mask, uint32_t seed, int len, uint8_t* out);
void   orc_cc_encode_k7(const uint8_t* bits, size_t n, uint8_t* out /* 2n */);

/* ---------- chains (reference hier blocks) ---------- */
typedef struct {
    size_t n_filtered, n_const, n_bits_a, n_bits_b;
    cf32* filtered;     /* port 0 */
    cf32* constellation;/* port 1 */
    uint8_t* bits_a;    /* port 2 */
    uint8_t* bits_b;    /* port 3 (2-branch modes) */
} orc_demod_out;
void orc_demod_out_free(orc_demod_out* o);

/* RX front end of gr_demod_base: rotator (+ decimator when fs >= 2 Msps). Returns count. */
size_t orc_frontend(const cf32* in, size_t n, int samp_rate, double carrier_offset_hz, cf32* out);
int    orc_frontend_taps(int samp_rate, float* taps); /* count; 0 if no resampler */

void orc_demod_2fsk(const cf32* in, size_t n, int sps, int samp_rate, int carrier_freq, int filter_width, int fm, orc_demod_out* o);
void orc_demod_gmsk(const cf32* in, size_t n, int sps, int samp_rate, int carrier_freq, int filter_width, orc_demod_out* o);
void orc_demod_qpsk(const cf32* in, size_t n, int sps, int samp_rate, int carrier_freq, int filter_width, orc_demod_out* o);

/* modulators: packed bytes in -> cf32 @ samp_rate out; returns sample count (out may be NULL to size) */
size_t orc_mod_2fsk(const uint8_t* bytes, size_t nbytes, int sps, int samp_rate, int carrier_freq, int filter_width, int fm, cf32* out);
size_t orc_mod_gmsk(const uint8_t* bytes, size_t nbytes, int sps, int samp_rate, int carrier_freq, int filter_width, cf32* out);
size_t orc_mod_qpsk(const uint8_t* bytes, size_t nbytes, int sps, int samp_rate, int carrier_freq, int filter_width, cf32* out);
/* TX back end of gr_mod_base: interpolate 1 Msps -> fs with low_pass(I, fs, 480k, 20k, BH) */
size_t orc_tx_interp(const cf32* in, size_t n, int samp_rate, cf32* out);

void orc_demod_4fsk(const cf32* in, size_t n, int sps, int samp_rate, int carrier_freq, int filter_width, int fm, orc_demod_out* o);
void orc_demod_bpsk(const cf32* in, size_t n, int sps, int samp_rate, int carrier_freq, int filter_width, orc_demod_out* o);
void   orc_preemph_taps(int sample_rate, double tau, double a[2], double b[2]);
size_t orc_mod_nbfm(const float* audio, size_t n, int sps, int samp_rate, int filter_width, float bb_gain, cf32* out);   /* out NULL: count */
size_t orc_mod_4fsk(const uint8_t* bytes, size_t nbytes, int sps, int samp_rate, int carrier_freq, int filter_width, int fm, cf32* out);
size_t orc_mod_m17(const uint8_t* bytes, size_t nbytes, int sps, int samp_rate, int filter_width, float bb_gain, cf32* out);
/* gr_mod_dmr (src/gr/gr_mod_dmr.cpp:26-90); zero_runs = {ignored, T, count} triples at the zero-idle block's 24 ksps input (NULL / 0: none) */
size_t orc_mod_dmr(const uint8_t* bytes, size_t nbytes, int sps, int samp_rate, int filter_width, float bb_gain, const uint64_t* zero_runs, size_t nruns, cf32* out);
void   orc_zero_idle_bursts_delay(const cf32* in, size_t n, unsigned delay, const uint64_t* runs, size_t nruns, cf32* out);   /* gr_zero_idle_bursts(delay > 0): history shift + tags matched `delay` items early, one work() call */
size_t orc_mod_dsss(const uint8_t* bytes, size_t nbytes, int sps, int samp_rate, int filter_width, float bb_gain, cf32* out);
size_t orc_mod_bpsk(const uint8_t* bytes, size_t nbytes, int sps, int samp_rate, int carrier_freq, int filter_width, cf32* out);
size_t orc_clock_recovery_mm_cc(const cf32* in, size_t n, float omega, float gain_omega, float mu, float gain_mu,
                                float omega_relative_limit, cf32* out);
int orc_modem_sync_geometry(int modem_type, int* bit_buf_len, int* frame_length);
size_t orc_modem_sync_collected(void);   /* of the last orc_modem_sync call on this thread's library state (tests are single threaded) */
size_t orc_modem_sync(int modem_type, const uint8_t* bits, size_t n, uint32_t st[5], uint8_t* bitbuf, uint8_t* out);
size_t orc_deframer(int type, const uint8_t* bits, size_t n, uint32_t st[3], uint8_t* out);
size_t orc_rssi_tag(const cf32* in, size_t n, float calibration, float* db);
size_t orc_demod_mmdvm(const cf32* in, size_t n, int samp_rate, int filter_width, int16_t* out, size_t cap, float* rssi, float cal,
                       size_t* n_rssi);
size_t orc_mod_mmdvm(const int16_t* in, size_t n, int filter_width, float bb_gain, cf32* out);
size_t orc_mod_mmdvm_multi(const int16_t* in, size_t n, int N, int filter_width, cf32* out);
void   orc_zero_idle_bursts(const cf32* in, size_t n, const uint64_t* runs, size_t nruns, cf32* out);   /* gr_zero_idle_bursts alone */
void   orc_set_zero_runs(const uint64_t* runs /* {channel, start, count} triples */, size_t nruns);   /* gr_zero_idle_bursts for the next orc_mod_mmdvm* call */
size_t orc_demod_mmdvm_xlating(const cf32* in, size_t n, int N, int separation, int D, int fw, int16_t* out, size_t cap,
                               float* rssi, size_t rcap, float cal);
size_t orc_demod_mmdvm_multi_4fsk(const cf32* in, size_t n, int M, int16_t* out, size_t cap, float* rssi, size_t rcap, float cal,
                                  uint8_t* dibits, size_t dcap, size_t* ndib);
size_t orc_demod_mmdvm_multi_rssi(const cf32* in, size_t n, int M, int16_t* out, size_t cap, float* rssi, size_t rcap, float cal);
size_t orc_mmdvm_channel_tails(const cf32* ch, int nch, size_t n1, int16_t* out, size_t cap, float* rssi, size_t rcap, float cal,
                               uint8_t* dibits, size_t dcap, size_t* ndib);   /* per-channel chains only (channel-sharded jobs) */
size_t orc_demod_mmdvm_xlating_bank_4fsk(const cf32* in, size_t n, int N, int16_t* out, size_t cap, float* rssi, size_t rcap, float cal,
                                         uint8_t* dibits, size_t dcap, size_t* ndib);   /* BASELINE configs[3] literal: N freq-xlating FIRs 1:N */
void orc_demod_dmr(const cf32* in, size_t n, int sps, int samp_rate, orc_demod_out* o);
void orc_demod_m17(const cf32* in, size_t n, int samp_rate, int filter_width, orc_demod_out* o);
int orc_dsss_taps(int sps, float* taps);
size_t orc_dsss_decoder(const cf32* in, size_t n, int sps, cf32* out);
void orc_demod_dsss(const cf32* in, size_t n, int sps, int samp_rate, int filter_width, orc_demod_out* o);
size_t orc_demod_dmr_port3(const cf32* in, size_t n, int samp_rate, float* out);
void orc_4fsk_symbols_to_bits(const float* sym, size_t nsym, cf32* constellation, uint8_t* bits);
/* multi-carrier MMDVM RX (gr_demod_mmdvm_multi2): PFB channelizer + per-channel 24/25 resampler, LPF, FM discriminator, int16 */
int    orc_chan_proto_taps(int M, float* taps);
void orc_chan_twiddles(int M, cf32* W);   /* exactly conjugate-symmetric DFT twiddles (channelizer contract) */
size_t orc_pfb_channelizer(const cf32* in, size_t n, const float* taps, int nt, int M, cf32* out);
size_t orc_demod_mmdvm_multi(const cf32* in, size_t n, int M, int16_t* out, size_t cap);

/* DMR DMO correlator slicer (gr_dmr_dmo_sink), orc_dmr.c.  Frames leave as 40-byte records {type, FN, colour code, 0, 33 bytes, 3 pad}. */
typedef struct {
    uint32_t bitBuffer[5];
    uint16_t bitPtr, dataPtr, syncPtr, startPtr, endPtr;
    float maxCorr, centre[4], threshold[4];
    uint8_t averagePtr, syncCount, state, control, n, colorCode;
    float buffer[1440];
} orc_dmo_state;
void     orc_dmo_init(orc_dmo_state* s);
size_t   orc_dmo_process(orc_dmo_state* s, const uint32_t* golay_table, const float* in, size_t n, uint8_t* out, size_t cap_frames);
uint32_t orc_golay1987_syndrome(uint32_t pattern);
void     orc_golay1987_table(uint32_t* table /* 2048 entries */);

/* batch drivers (OpenMP over streams) used by the cpu_baseline leg of bench.py */
enum { ORC_MODE_2FSK_1K = 0, ORC_MODE_GMSK_10K = 1, ORC_MODE_QPSK_250K = 2 };
double orc_batch_rx(int mode, const cf32* iq, int batch, size_t n, int samp_rate, double carrier_offset_hz,
                    int threads, uint64_t* bit_checksum);

#ifdef __cplusplus
}
/* ---- construction trace (orc_trace.c): which primitives a chain calls, with which parameters; tests/test_ref_chains.py ---- */
void orc_trace_enable(int on);            /* clears the trace; on != 0 starts recording */
int  orc_trace_on(void);
const char* orc_trace_get(void);          /* one "name(args)" line per primitive call */
void orc_trace_event(const char* fmt, ...);
void orc_trace_taps(const void* taps, size_t bytes, const char* fmt, ...);
const char* orc_trace_name(const void* taps, size_t bytes);

#endif
/* side outputs of gr_demod_base (orc_side.c) */
/* analogue voice receivers (orc_analog.c) */
int    orc_complex_band_pass_2(double gain, double fs, double lo, double hi, double tw, double atten_db, int win, cf32* taps);
void   orc_deemph_taps(int sample_rate, double tau, double a[2], double b[2]);
void   orc_squelch_envelope(int ramp, float* env /* ramp + 1 */);
size_t orc_pwr_squelch_cc(const cf32* in, size_t n, double db, double alpha, int ramp, int gate, cf32* out);
void   orc_agc2_ff(const float* in, size_t n, float attack, float decay, float ref, float gain, float max_gain, float* out);
void   orc_iir_ffd_2(const float* in, size_t n, const double ff[2], const double fb[2], int oldstyle, float* out);
void   orc_set_tx_ctcss(float tone_hz);   /* gr_mod_nbfm::set_ctcss for the next orc_mod_nbfm calls: > 0 tone on, < 0 switched off again (x0.98), 0 = constructor */
void   orc_fxpt_sine_table(float* tab /* 1024 x 2 */); uint32_t orc_fxpt_phase_inc(double fs, double freq);
void   orc_sig_source_sin(double fs, double freq, double ampl, float offset, uint64_t k0, size_t n, float* out);   /* analog::sig_source_f(GR_SIN_WAVE, offset) [GR-MEM]: the CW key's tone */
void   orc_sig_source_cos(double fs, double freq, double ampl, uint64_t k0, size_t n, float* out);   /* analog::sig_source_f(GR_COS_WAVE) [GR-MEM] */
void   orc_set_rx_filter_width(int width);   /* gr_demod_nbfm / am / wbfm / ssb::set_filter_width for the next orc_demod_analog / orc_demod_ssb calls; 0 = constructor */
void   orc_set_rx_gain(float k);             /* gr_demod_ssb::set_gain (_if_gain) for the next orc_demod_ssb calls; < 0 = the constructor's 0.9 */
void   orc_set_tx_filter_width(int width);   /* gr_mod_nbfm / am / ssb::set_filter_width for the next orc_mod_nbfm / orc_mod_am / orc_mod_ssb calls; 0 = constructor */
void   orc_set_ctcss(float tone_hz);   /* gr_demod_nbfm::set_ctcss for the next orc_demod_analog(kind 0) calls; 0 = off */
size_t orc_ctcss_squelch_ff(const float* in, size_t n, int rate, float freq, double level, int len, int ramp, int gate, float* out);
void   orc_ctcss_freqs(float freq, float* f_l, float* f_r);
void   orc_goertzel_coeffs(int rate, float freq, float* wr, float* wi);
void   orc_demod_analog(const cf32* in, size_t n, int kind /* 0 NBFM, 1 AM, 2 WBFM */, int samp_rate, int filter_width,
                        cf32** filtered, size_t* n_filtered, float** audio, size_t* n_audio);
int    orc_band_pass_2(double gain, double fs, double lo, double hi, double tw, double atten_db, int win, float* taps);
void   orc_cessb_clipper(const cf32* in, size_t n, float clip, cf32* out);
size_t orc_cessb_stretcher(const cf32* in, size_t n, cf32* out);
void   orc_demod_ssb(const cf32* in, size_t n, int samp_rate, int filter_width, int sb /* 0 USB, 1 LSB */,
                     cf32** filtered, size_t* n_filtered, float** audio, size_t* n_audio);
size_t orc_mod_ssb(const float* audio, size_t n, int sps, int samp_rate, int filter_width, int sb, float bb_gain, cf32* out);   /* out NULL: count */
size_t orc_mod_am(const float* audio, size_t n, int sps, int samp_rate, int filter_width, float bb_gain, cf32* out);           /* gr_mod_am; out NULL: count */
void   orc_free(void* p);
/* frame FEC of the DMR / M17 stacks (orc_framefec.c; PINNED against the real reference sources through oracle/_ref) */
void     orc_bptc19696_decode(const uint8_t* in33, uint8_t* out12);
void     orc_bptc19696_encode(const uint8_t* in12, uint8_t* inout33);
uint32_t orc_golay24_encode(uint16_t data);
uint16_t orc_golay24_decode(uint32_t codeword);
const uint8_t* orc_m17_sequence(void);   /* 46 bytes */
void     orc_m17_decode_frame(const uint8_t frame[48], uint8_t rec[40]);
void     orc_m17_encode_frame(const uint8_t rec[40], uint8_t frame[48]);
uint16_t orc_m17_crc16(const uint8_t* p, size_t n);
float orc_det_log2f(float x);
void orc_rssi_block(const cf32* in, size_t n, float level, float* out);
void orc_power_spectrum(const cf32* in, const float* window, size_t n, float* out);

/* ---- construction trace (orc_trace.c): which primitives a chain calls, with which parameters; tests/test_ref_chains.py ---- */
void orc_trace_enable(int on);            /* clears the trace; on != 0 starts recording */
int  orc_trace_on(void);
const char* orc_trace_get(void);          /* one "name(args)" line per primitive call */
void orc_trace_event(const char* fmt, ...);
void orc_trace_taps(const void* taps, size_t bytes, const char* fmt, ...);
const char* orc_trace_name(const void* taps, size_t bytes);

/* orc_pipeline.c: SURVEY 8(d) CPU baseline variant (ii), the thread-per-block scheduler EMULATED -- one thread per block of the 2FSK-1k
 * receiver at 1 Msps, bounded queues in between; returns wall seconds, the checksum orc_batch_rx gives for the same input, busy[11] */
double orc_pipeline_rx_2fsk1k(const cf32* iq, int batch, size_t n, double carrier_offset_hz, uint64_t* bit_checksum, double* busy, int* nstages);

#endif
