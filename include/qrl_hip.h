/*
 * qrl_hip.h — C ABI of libqrl_hip.so, the MI355X-native drop-in for QRadioLink's gr_modem
 * RX DSP hot path (the per-mode demod flowgraphs under src/gr/ of the reference).
 *
 * Nothing like this exists in the reference: there the path is a GNU Radio flowgraph
 * (gr::top_block "demodulator", src/gr/gr_demod_base.cpp:32) whose blocks run on CPU threads.
 * Each entry point below names the reference interface it replaces; INTEGRATION.md shows the
 * C++ stub a maintainer would add inside gr_demod_base / a gr::sync_block to bind it.
 *
 * Rules: plain C types; no exceptions cross the ABI; every function returns QRL_OK (0) or a
 * negative qrl_status; the caller owns every buffer it passes, the library owns device state.
 * One handle is single-threaded (like one GNU Radio block: work() calls never overlap,
 * SURVEY.md 8b); different handles may be driven concurrently on different HIP streams.
 *
 * Data layout: "cf32" = interleaved complex<float> (gr_complex, 8 bytes). A demodulator
 * handle processes `batch` independent streams per call: iq[b*stride + i], i < n.
 * All data pointers are DEVICE pointers unless the function name ends in _host.
 */
#ifndef QRL_HIP_H
#define QRL_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    QRL_OK = 0,
    QRL_ERR_ARG = -1,        /* invalid argument / unsupported mode */
    QRL_ERR_NO_DEVICE = -2,  /* no HIP device / wrong architecture */
    QRL_ERR_HIP = -3,        /* a HIP runtime call failed (see qrl_last_error) */
    QRL_ERR_NOMEM = -4,
    QRL_ERR_TOO_BIG = -5,    /* n exceeds max_chunk given at creation */
    QRL_ERR_STATE = -6
} qrl_status;

/* values of gr_modem_types (reference src/modem_types.h:5-50) accepted by qrl_demod_create */
enum {
    QRL_MODEM_BPSK2K = 0, QRL_MODEM_QPSK20K = 1, QRL_MODEM_QPSKVIDEO = 2, QRL_MODEM_4FSK2K = 3, QRL_MODEM_4FSK10KFM = 4, QRL_MODEM_4FSK2KFM = 5, QRL_MODEM_4FSK1KFM = 6, QRL_MODEM_QPSK2K = 7,
    QRL_MODEM_NBFM2500 = 8, QRL_MODEM_NBFM5000 = 9, QRL_MODEM_WBFM = 10, QRL_MODEM_USB2500 = 11, QRL_MODEM_LSB2500 = 12, QRL_MODEM_CW600USB = 13 /* TX only: qrl_amod */, QRL_MODEM_AM5000 = 14,   /* analogue voice receivers: port 1 = audio */
    QRL_MODEM_2FSK2KFM = 15, QRL_MODEM_2FSK1KFM = 16, QRL_MODEM_2FSK2K = 17, QRL_MODEM_2FSK1K = 18,
    QRL_MODEM_2FSK10KFM = 19, QRL_MODEM_GMSK2K = 20, QRL_MODEM_GMSK1K = 21, QRL_MODEM_GMSK10K = 22,
    QRL_MODEM_BPSK1K = 24, QRL_MODEM_BPSK8 = 25 /* DSSS, Barker 13 */, QRL_MODEM_QPSK250K = 26, QRL_MODEM_4FSK100K = 27, QRL_MODEM_M17 = 40, QRL_MODEM_DMR = 41
};

typedef struct qrl_ctx qrl_ctx;
typedef struct qrl_demod qrl_demod;

/* replaces: process-level GNU Radio/VOLK initialisation (gr::make_top_block, gr_demod_base.cpp:32).
 * Fails with QRL_ERR_NO_DEVICE when no gfx950-class HIP device is usable: there is no CPU fallback. */
int qrl_init(int device, qrl_ctx** ctx);
void qrl_shutdown(qrl_ctx* ctx);
const char* qrl_strerror(int status);
const char* qrl_last_error(void);
/* compile-time identification of the library (no GPU needed) */
const char* qrl_version(void);

/*
 * Configuration of one RX chain = the constructor arguments the reference passes.
 * replaces: make_gr_demod_2fsk(sps,samp_rate,carrier_freq,filter_width,fm) gr_demod_2fsk.cpp:19-37,
 *           make_gr_demod_gmsk(...) gr_demod_gmsk.cpp:19-37, make_gr_demod_qpsk(...) gr_demod_qpsk.cpp:20-37,
 *           make_gr_demod_4fsk(..., fm) gr_demod_4fsk.cpp:19-36, make_gr_demod_bpsk(...) gr_demod_bpsk.cpp:19-35,
 *           make_gr_demod_dmr(sps, samp_rate) gr_demod_dmr.cpp:19-35
 *           with the literals of gr_demod_base.cpp:203-253 when use_mode_defaults != 0;
 *           gr_demod_base::set_samp_rate (gr_demod_base.cpp:1303-1362) via device_samp_rate;
 *           gr_demod_base::set_carrier_offset (gr_demod_base.cpp:1220-1225) via carrier_offset_hz.
 */
typedef struct {
    int modem_type;          /* gr_modem_types value */
    int use_mode_defaults;   /* 1: take sps/filter_width/fm from the reference's table for modem_type */
    int sps, samp_rate, carrier_freq, filter_width, fm; /* make_gr_demod_X args (samp_rate = 1000000) */
    int device_samp_rate;    /* SDR rate; >= 2e6 inserts rotator + 1:fs/1e6 decimator, else rotator only */
    double carrier_offset_hz;
    int batch;               /* independent streams per call (>= 1) */
    size_t max_chunk;        /* largest n (samples per stream) of any process call */
    void* hip_stream;        /* hipStream_t to run on, NULL = the library creates one */
    int enable_side_outputs; /* 1: also produce port 0 (filtered) and port 1 (constellation) */
    /* the time-domain scope tap's resampler (qrl_demod_set_time_domain_output), 0 = the constructor's 1:10 with low_pass(1, 1e6, 50000, 25000, HAMMING):
     * time_domain_samp_rate replaces gr_demod_base::set_time_sink_samp_rate(samp_rate) (src/gr/gr_demod_base.cpp:1249-1290): decimation 1e6 / samp_rate
     * (integer), taps low_pass(1, 1e6, samp_rate / 2 - samp_rate / 8, samp_rate / 4, HAMMING) (integer divisions as in the reference);
     * time_domain_filter_width > 0 replaces gr_demod_base::set_time_domain_filter_width(width) (:1292-1301) called after it: taps low_pass(1, 1e6, width, width,
     * HAMMING), same decimation.  Create-time values: the history a handle keeps depends on the filter (the host facade re-opens its handle, the reference
     * re-creates the block under lock()). */
    int time_domain_samp_rate;
    double time_domain_filter_width;
} qrl_demod_config;

/* per-call outputs, all caller-allocated DEVICE buffers laid out [batch][cap];
 * counts[b*4 + k] = items written for stream b on port k (0 filtered, 1 constellation, 2 bits A, 3 bits B).
 * A NULL data pointer skips that port (its count is still reported). */
typedef struct {
    float* filtered;      size_t filtered_cap;      /* cf32, port 0: gr_demod_2fsk.cpp:133 */
    float* constellation; size_t constellation_cap; /* cf32, port 1: gr_demod_2fsk.cpp:153-154 */
    uint8_t* bits_a;      size_t bits_cap;          /* u8 one bit per byte, port 2 */
    uint8_t* bits_b;                                /* port 3 (two-branch modes), same cap */
    uint32_t* counts;                               /* [batch][4] */
    float* audio;         size_t audio_cap;         /* analogue modes: port 1 = f32 audio at 8 ksps (counts[b][1] = samples of the call);
                                                       written whether or not side outputs are enabled */
} qrl_demod_out;

int qrl_demod_create(qrl_ctx* ctx, const qrl_demod_config* cfg, qrl_demod** out);
void qrl_demod_destroy(qrl_demod* d);
/* replaces: top_block lock/flush on set_mode (gr_demod_base.cpp:302-311): zero all DSP state */
int qrl_demod_reset(qrl_demod* d);
/* replaces: rotator_cc::set_phase_inc (gr_demod_base.cpp:1220-1225); phase-continuous retune */
int qrl_demod_set_carrier_offset(qrl_demod* d, double carrier_offset_hz);
/* capacities (items per stream) a call with n input samples can need */
/* a37b, QRL_MODEM_DMR only: gr_dmr_dmo_sink (reference src/gr/gr_dmr_dmo_sink.cpp:63-357) on the device, fed from port 3 of
 * gr_demod_dmr (RRC-filtered discriminator output, gr_demod_dmr.cpp:94).  Every following qrl_demod_process call also runs the
 * MS-sync correlator / 4-level slicer / slot-type state machine over the call's new samples and appends the DMR bursts it cuts to
 * frames[b * cap_frames * 40 + i * 40]: 40-byte records {frame type (DMRFrameType 0 data, 1 voice, 2 voice sync), FN, colour code,
 * 0, 33 frame bytes, 3 pad}; counts[b] = bursts found by the call (device pointers; replaces gr_dmr_dmo_sink::get_data); only the first
 * cap_frames records are written, so counts[b] > cap_frames means bursts were dropped (one burst lasts 30 ms: size cap_frames for the chunk).
 * frames == NULL switches it off.  The stream position must be the start of the stream (call it before the first process / after
 * a reset) or the slicer starts with an empty 1440-sample history. */
#define QRL_DMO_RECORD_BYTES 40
int qrl_demod_set_dmo_output(qrl_demod* d, uint8_t* frames, size_t cap_frames, uint32_t* counts);
/* Per-handle run-time options (none of them changes results; tests/test_gpu_parity.py checks each value of each option against the oracle).  QRL_OPT_OVERLAP (2FSK family only, default 1 since round 3): value 1 runs
 * everything behind the first decimated ring of call k on a second stream under the front end of call k + 1; value 0 runs the
 * kernels of a call one after another.  QRL_OPT_UNFUSED_DEC2 (QPSK chains with sps <= 4, default 0): value 1 runs the 1:2 resampler
 * and the shaping filter (gr_demod_qpsk.cpp:92-103) as the two kernels of rounds 1-2 instead of the fused one (A/B and parity checks);
 * only before the first sample of a stream.  QRL_OPT_GROUPED (gr_demod_qpsk chain; default: 1 when the batch gives the recursion kernel a
 * workgroup for at least every second CU, else 0): the order in which a call's three kernels are put on the device -- value 1: front
 * end of call k + 1 behind the recursion of call k, the decoder of call k - 1 launched with the recursion of call k (flushed by every
 * function that waits for results); value 0: three free-running streams.  QRL_OPT_INPUT_RESIDENT (default 0; a handle that owns its
 * streams): value 1 is the caller's promise that the IQ of every call is COMPLETE in device memory when qrl_demod_process is called -- not
 * the product of work the caller has queued on qrl_demod_stream() and not yet waited for.  The two helper kernels of the front end
 * (the history kept for the next call, the edge scratch of this one) then read it on a fourth internal stream beside the front end of the
 * call before instead of in line with it; qrl_demod_sync / qrl_demod_stream_wait cover that stream too.  With value 0 everything that reads
 * the caller's buffer is ordered behind the handle's stream, as qrl_demod_process documents. */
enum { QRL_OPT_OVERLAP = 1, QRL_OPT_UNFUSED_DEC2 = 2, QRL_OPT_FLL_SLIM = 3 /* tuning: single-wave FLL workgroups */, QRL_OPT_GROUPED = 4, QRL_OPT_INPUT_RESIDENT = 5 };
int qrl_demod_set_option(qrl_demod* d, int option, int value);
int qrl_demod_out_caps(const qrl_demod* d, size_t n, size_t* filtered_cap, size_t* constellation_cap, size_t* bits_cap);
/* analogue voice receivers (QRL_MODEM_NBFM2500 / NBFM5000 / AM5000 / WBFM / USB2500 / LSB2500; replace make_gr_demod_nbfm / _am /
 * _wbfm / _ssb, reference src/gr/gr_demod_nbfm.cpp:19-88, gr_demod_am.cpp:19-79, gr_demod_wbfm.cpp:19-72, gr_demod_ssb.cpp:19-81
 * (with src/gr/cessb/clipper_cc_impl.cc, stretcher_cc_impl.cc), instances gr_demod_base.cpp:215,219,220,226-228):
 * port 0 = the channel-filtered IQ, port 1 = audio.  The squelch gates (pwr_squelch_cc(-140, 0.01, ramp, true)), so the number of
 * audio samples a call returns depends on the signal; audio_cap(n) is the bound for a call of n input samples (0 for other modes). */
int qrl_demod_audio_cap(const qrl_demod* d, size_t n, size_t* audio_cap);
/* replaces gr_demod_nbfm/am/wbfm::set_squelch (pwr_squelch_cc::set_threshold, gr_demod_base.cpp:1186-1199): threshold in dB */
int qrl_demod_set_squelch(qrl_demod* d, double db);
/* replaces the time-domain scope tap of gr_demod_base: enable_time_domain (src/gr/gr_demod_base.cpp:1115-1147) connects _demod_valve ->
 * rational_resampler_ccf(1, 10, low_pass(1, 1e6, 50000, 25000, HAMMING)) (:62-63) -> gr_sample_sink (src/gr/gr_sample_sink.cpp), read
 * through get_sample_data (:988-1013).  samples != NULL switches the tap on: every following qrl_demod_process also decimates the
 * 1 Msps signal behind the front end (the caller's rotated IQ when the device runs at 1 Msps) 1:10 and writes the call's new
 * 100 ksps items to samples[b*cap + k] (cf32 pairs, device memory), counts[b] = items written; NULL switches it off.
 * qrl_demod_time_domain_cap: the bound for a call of n input samples.  The sink's mailbox rules (window, drop threshold) are host
 * logic: qrl_host::gr_demod_base_hip::get_sample_data. */
int qrl_demod_time_domain_cap(const qrl_demod* d, size_t n, size_t* cap);
int qrl_demod_set_time_domain_output(qrl_demod* d, float* samples, size_t cap, uint32_t* counts);
/* replaces gr_demod_nbfm::set_ctcss(value) (src/gr/gr_demod_nbfm.cpp:97-123): tone_hz != 0 switches analog::ctcss_squelch_ff(8000, tone,
 * 0.01, 8000, 160, true) (:59-60) in between the audio resampler and the audio filter and the audio filter to band_pass_2(1, 8000,
 * 300, 3500, 200, 35, BH); 0 switches it out again (the constructor's graph).  Switching it in or out restarts the chain from a
 * fresh state (the reference re-wires the graph under lock()); a different tone while it is on re-initialises the tone detector
 * only (ctcss_squelch_ff::set_frequency).  NBFM receivers only.  The block gates: audio leaves only while the tone is present. */
int qrl_demod_set_ctcss(qrl_demod* d, float tone_hz);
/* replaces gr_demod_am / gr_demod_ssb::set_agc_attack / set_agc_decay (gr_demod_am.cpp:90-98, gr_demod_ssb.cpp:104-112) */
int qrl_demod_set_agc(qrl_demod* d, float attack, float decay);
/* replaces gr_demod_base::set_filter_width(filter_width, mode) (src/gr/gr_demod_base.cpp:1155-1185), which forwards to the analogue receiver of
 * `mode`: gr_demod_nbfm::set_filter_width (gr_demod_nbfm.cpp:82-90), gr_demod_am (gr_demod_am.cpp:84-91), gr_demod_wbfm (gr_demod_wbfm.cpp:76-84),
 * gr_demod_ssb (gr_demod_ssb.cpp:89-101).  The setters do NOT repeat the constructors' designs: NBFM / WBFM low_pass(1, fs, w, 1200, BH) and the
 * discriminator gain fs / (4 pi w) resp. fs / (2 pi w); AM complex_band_pass(1, fs, -w, w, 1200, BH); SSB the constructor's band-pass with w and the
 * audio filter band_pass_2(2, 8000, 200, w, 200, 90, BH) -- gain 2.  The reference swaps the taps of a running graph at a sample its scheduler decides;
 * here the chain restarts from a fresh state (like qrl_demod_reset; squelch / CTCSS / AGC settings are kept).  Analogue receivers only. */
int qrl_demod_set_filter_width(qrl_demod* d, int filter_width);
/* replaces gr_demod_base::set_gain(value) (src/gr/gr_demod_base.cpp:1206-1210) -> gr_demod_ssb::set_gain = _if_gain->set_k(value) (gr_demod_ssb.cpp:118-121;
 * constructor: 0.9).  Takes effect with the next qrl_demod_process call.  SSB receivers only. */
int qrl_demod_set_gain(qrl_demod* d, float value);

/* replaces: one scheduler pass of the "demodulator" top_block over n new samples per stream:
 * (the serial tail of a call runs on an internal second HIP stream and overlaps the next call; only
 * qrl_demod_sync() -- not a sync of the handle's own stream -- guarantees the outputs are complete)
 * gr::sync_block::work()/general_work() of every block on the chain (SURVEY.md 8b block ABI).
 * iq: device pointer, stream b at iq + 2*b*stride floats, n <= max_chunk samples each, base and
 * stride*8 bytes 16-byte aligned.  Asynchronous on the handle's stream; results are valid after
 * qrl_demod_sync().  Results are independent of how the stream is cut into calls.
 * Calls in flight together need DISTINCT output buffers: for the QPSK / BPSK / 4FSK-discriminator families the recursion and the
 * Viterbi decoder of a call run on internal streams up to two calls behind the front end, so a caller that reuses one qrl_demod_out
 * for back-to-back calls must put qrl_demod_sync() or qrl_demod_stream_wait() between them (or alternate two buffer sets). */
int qrl_demod_process(qrl_demod* d, const float* iq, size_t stride, size_t n, const qrl_demod_out* out);
int qrl_demod_sync(qrl_demod* d);
void* qrl_demod_stream(qrl_demod* d); /* hipStream_t */
/* Profiling aid: the HIP streams a call's stages are launched on -- out[0] the handle's stream (front end), out[1] the stream of the
 * decimated-rate / recursion kernels, out[2] the decoder's.  An event recorded on one of them orders nothing (unlike
 * qrl_demod_stream_wait, whose wait on the caller's stream sits in a hardware queue that stream may share with this handle's);
 * bench.py takes its per-step completion times this way.  Not a synchronisation interface: use qrl_demod_sync / _stream_wait. */
int qrl_demod_internal_streams(qrl_demod* d, void* out[3]);
/* makes the caller's stream wait (on the device, without blocking the host) for everything this handle has enqueued so far on its
 * internal streams: how a host layer chains its own asynchronous copies of the output ports behind a qrl_demod_process call
 * (host/gr_modem_hip.cpp: double-buffered mailboxes) instead of calling qrl_demod_sync. */
int qrl_demod_stream_wait(qrl_demod* d, void* hip_stream);

/* Per-kernel timing of the dominant (HBM-facing) kernel with HIP events recorded on the handle's own
 * stream (bench.py roofline leg).  enable!=0 starts recording one event pair per process call;
 * qrl_demod_profile_read syncs, returns the summed duration and launch count, and clears the record. */
int qrl_demod_profile(qrl_demod* d, int enable);
int qrl_demod_profile_read(qrl_demod* d, double* kernel_ms, uint64_t* launches, const char** kernel_name);

/* host-buffer convenience used by the C++ adaptor (gr_bit_sink-style mailboxes): copies iq H2D,
 * runs one pass, copies bits back.  bits_x_host: [batch][bits_cap]; counts_host: [batch][4]. */
int qrl_demod_process_host(qrl_demod* d, const float* iq_host, size_t stride, size_t n,
                           uint8_t* bits_a_host, uint8_t* bits_b_host, size_t bits_cap, uint32_t* counts_host);

/* ---- TX: the "modulator" top_block (reference src/gr/gr_mod_base.cpp:25) ----------------------------------
 * One handle = make_gr_mod_qpsk(sps, samp_rate, carrier_freq, filter_width) (src/gr/gr_mod_qpsk.cpp:19-30;
 * instance make_gr_mod_qpsk(4,1000000,1700,160000) src/gr/gr_mod_base.cpp:175) for `batch` independent streams.
 * Built: QPSK 250k / video / 20k / 2k (gr_mod_qpsk.cpp), 2FSK incl. FM variants (gr_mod_2fsk.cpp:19-99), GMSK
 * (gr_mod_gmsk.cpp:19-95), 4FSK incl. the non-FM 2k mode (gr_mod_4fsk.cpp:19-115), BPSK (gr_mod_bpsk.cpp:19-67), and
 * gr_mod_base's rotator + rate-matching interpolator (device_samp_rate / carrier_offset_hz below). */
typedef struct qrl_mod qrl_mod;
typedef struct {
    int modem_type;          /* gr_modem_types value (QRL_MODEM_QPSK250K) */
    int use_mode_defaults;   /* 1: sps/filter_width from gr_mod_base.cpp:175 */
    int sps, samp_rate, carrier_freq, filter_width, fm;   /* make_gr_mod_qpsk / make_gr_mod_2fsk / make_gr_mod_gmsk arguments */
    int batch;               /* independent streams per call */
    size_t max_bytes;        /* largest nbytes of any process call */
    void* hip_stream;        /* hipStream_t, NULL = the library creates one (RX and TX handles on different streams run
                                concurrently: full duplex, reference src/radiocontroller.cpp:2043-2078) */
    float bb_gain;           /* gr_mod_qpsk::set_bb_gain, 0 = 1.0 */
    /* gr_mod_base back end (src/gr/gr_mod_base.cpp:38,215-258): modulator (1 Msps) -> rotator_cc(2 pi offset / 1e6) ->
     * rational_resampler_ccf(fs/1e6, 1, low_pass(I, fs, 480k, 20k, BLACKMAN_HARRIS)) when fs >= 2 Msps -> SDR sink */
    int device_samp_rate;    /* gr_mod_base::set_samp_rate: 0 / 1e6 = no resampler, else a multiple of 1e6 in [2e6, 64e6] */
    double carrier_offset_hz;/* gr_mod_base::set_carrier_offset (rotator at 1 Msps) */
} qrl_mod_config;
int qrl_mod_create(qrl_ctx* ctx, const qrl_mod_config* cfg, qrl_mod** out);
void qrl_mod_destroy(qrl_mod* m);
/* replaces: flush of the modulator graph on mode change (gr_mod_base.cpp:354-360): scrambler seed, encoder, filter history */
int qrl_mod_reset(qrl_mod* m);
/* replaces: gr_mod_qpsk::set_bb_gain (src/gr/gr_mod_qpsk.cpp:91-94) */
int qrl_mod_set_bb_gain(qrl_mod* m, float value);
/* replaces: gr_mod_base::set_carrier_offset -> rotator_cc::set_phase_inc (src/gr/gr_mod_base.cpp:799-805); phase-continuous.
 * Only for handles created with the back end (device_samp_rate >= 2e6 or a non-zero initial offset). */
int qrl_mod_set_carrier_offset(qrl_mod* m, double hz);
size_t qrl_mod_samples_per_byte(const qrl_mod* m);
typedef struct { int stream; int channel; uint64_t start; uint64_t count; } qrl_zero_run;
/* QRL_MODEM_DMR (replaces make_gr_mod_dmr(), reference src/gr/gr_mod_dmr.cpp:19-90, instance gr_mod_base.cpp:207, fed by gr_mod_base::setDMRData
 * :788 -> gr_dmr_source): raw dibits -> map{2,3,1,0} -> 4 levels -> RRC(5, 24000, 4800, 0.2, 125) x5 -> x0.66666666 -> frequency_modulator_fc
 * (pi 4800 0.85 / 24000) -> gr_zero_idle_bursts(62) -> x0.9 -> bb gain -> rational_resampler_ccf(125, 3, low_pass_2(125, 3e6, 5000, 2000, 60, BH)):
 * 2500 samples per 3 bytes, like QRL_MODEM_M17.  The zero-idle block is restated as ONE work() call over the stream sees it: its history of
 * 2 x 720 items delays the signal by 1439 items at 24 ksps, and a "zero_samples" tag {offset T, count} zeroes the block's OUTPUT items
 * T - 62 ... T - 62 + count - 1 (a later tag ends an earlier run).  qrl_mod_add_zero_runs hands over those tags (T in the block's 24 ksps
 * input coordinates = 20 x the byte offset gr_dmr_source put the tag on; `channel` ignored); they apply to the following qrl_mod_process calls.
 * (Across scheduler calls the reference only matches tags inside a call's own window, so it misses tags within the first 62 items of a window:
 * not restated.)  The DMR frame / CACH / slot-timing layer that produces bytes and tags (gr_dmr_source, src/DMR/) stays with the caller. */
int qrl_mod_add_zero_runs(qrl_mod* m, const qrl_zero_run* runs, size_t n);
/* QRL_MODEM_M17 (replaces make_gr_mod_m17(), reference src/gr/gr_mod_m17.cpp:19-81, gr_mod_base.cpp:206): its 125 / 3 output resampler
 * makes 833 1/3 samples per byte, so calls take multiples of 3 bytes (an M17 frame is 48) and return 2500 samples per 3 bytes;
 * qrl_mod_samples_per_byte is 0 for it.  Other modes: bytes_per_block 1, the value of qrl_mod_samples_per_byte. */
size_t qrl_mod_samples_per_block(const qrl_mod* m, size_t* bytes_per_block);   /* at the DEVICE rate: 8*sps (QPSK) or 16*sps*interp (FSK), times fs/1e6 */
/* replaces: gr_mod_base::set_data (src/gr/gr_mod_base.cpp:783-786) + gr_byte_source::work (src/gr/gr_byte_source.cpp:75-106)
 * + one scheduler pass of every block of gr_mod_qpsk: bytes[b*stride + i], i < nbytes (device, packed, MSB first) ->
 * iq[2*(b*out_stride + k)], k < nbytes*8*sps (device cf32).  State (scrambler, encoder, differential symbol, pulse-shaping
 * history) carries across calls; asynchronous on the handle's stream. */
int qrl_mod_process(qrl_mod* m, const uint8_t* bytes, size_t stride, size_t nbytes, float* iq, size_t out_stride);
int qrl_mod_sync(qrl_mod* m);
void* qrl_mod_stream(qrl_mod* m);

/* ---- multi-carrier MMDVM receiver (reference src/gr/gr_demod_mmdvm_multi2.cpp:19-38,58-135) ---------------------
 * make_gr_demod_mmdvm_multi2(burst_timer, num_channels, channel_separation, use_tdma, sps, samp_rate, carrier_freq,
 * filter_width): stream_to_streams + pfb_channelizer_ccf + per channel {24/25 resampler, LPF, FM discriminator,
 * level, float_to_short}.  The reference fixes 10 branches at 250 ksps (src/config_mmdvm.h:4); here num_channels = M
 * (2..64) at fs = 25 kHz * M; num_channels = 1 selects the SINGLE-carrier receiver gr_demod_mmdvm (src/gr/gr_demod_mmdvm.cpp:29-61:
 * 250 ksps in, rational_resampler_ccf(12, 125), rssi tag, 43-tap LPF, quadrature_demod_cf(24000/(2 pi 10000)), int16).  Channel c is centred at +c*fs/M (c > M/2: negative offsets); the reference's port
 * order {0,1,2,3,9,8,7} and the TDMA tagging / ZeroMQ framing of gr_mmdvm_sink (src/gr/gr_mmdvm_sink.cpp:66-176)
 * stay with the caller.  channel_first/channel_count select the channels THIS handle produces: ranks of a multi-GPU job
 * take disjoint ranges (no data-path collective). */
typedef struct qrl_chan qrl_chan;
typedef struct {
    int num_channels;        /* M: PFB branches = channels on the fs/M grid */
    int channel_first, channel_count;   /* produced range; channel_count <= 0: all */
    int batch;               /* independent wideband inputs per call */
    size_t max_chunk;        /* largest n (wideband samples per input) of any call */
    void* hip_stream;
    /* form 1 = the legacy "freq-xlating" receiver make_gr_demod_mmdvm_multi(burst_timer, num_channels, channel_separation,
     * use_tdma, sps, samp_rate, carrier_freq, filter_width) (src/gr/gr_demod_mmdvm_multi.cpp:19-38,58-123): per channel
     * rotator_cc(2 pi (-channel_separation) ct / fs) -> rational_resampler_ccf(1, decimation, low_pass(1, fs, filter_width,
     * 3500, BH)) -> fft_filter_ccf -> rssi tag -> discriminator -> int16, fs = 24 kHz * decimation (240 ksps in the
     * reference).  The wideband input is read once per channel (compute-bound form); 0 = PFB channelizer (multi2).
     * form 2 = BASELINE.json configs[3] taken literally ("64x freq-xlating-FIR channelizer + 4FSK demod"): channel i
     * = rotator_cc(2 pi (-25000) ct / fs) -> rational_resampler_ccf(1, num_channels, low_pass_2(1, fs, 5000, 2000, 60, BH)) at
     * fs = 25 kHz * num_channels with the PFB form's channel map (ct = i, i <= N/2; i - N above), followed by the per-channel
     * chain of form 0 (24/25 resampler, LPF, RSSI, discriminator -> int16, optional 4FSK tail).  The same channels as form 0 from
     * num_channels separate FIRs: the compute-bound way of doing what the PFB does; it exists to be measured next to it.
     * form 3 = the per-channel chain alone (qrl_chan_process_channels): `batch` counts 25 ksps channel streams, max_chunk channel samples. */
    int form;
    int channel_separation;  /* form 1: Hz, 0 = 25000 */
    int decimation;          /* form 1: 0 = 10 */
    int filter_width;        /* form 1: 0 = 8000 (header default, gr_demod_mmdvm_multi.h:37-40) */
} qrl_chan_config;
int qrl_chan_create(qrl_ctx* ctx, const qrl_chan_config* cfg, qrl_chan** out);
void qrl_chan_destroy(qrl_chan* c);
int qrl_chan_reset(qrl_chan* c);
/* path selection for tests (results are identical): the 64-channel geometry runs on k_pfb_stream64 + the fused per-channel kernel; every other
 * shape (other channel counts, unaligned rows, the single-carrier and frequency-translating forms) on the general-M channelizer and the separate
 * per-channel kernels.  QRL_CHAN_OPT_LEGACY_PFB = 1 / QRL_CHAN_OPT_LEGACY_TAIL = 1 send the 64-channel geometry down those general paths so that
 * the parity tests cover them at the bench shape too (round 6 deleted the third channelizer, round 3's tiled k_pfb_chan64).  LEGACY_TAIL only before the first samples of a stream or after qrl_chan_reset (QRL_ERR_STATE
 * otherwise: the fused kernel does not fill the intermediate rings the separate kernels take their history from).
 * QRL_CHAN_OPT_SERIAL_TAIL = 1 keeps the per-channel kernel of a call on the handle's stream, behind its channelizer and in front of the
 * next one (rounds 1-4); 0, the default for a PFB-form handle that owns its stream: it runs on an internal stream BESIDE the channelizer
 * of the next call (the channel ring holds two calls) -- int16 / RSSI outputs are then valid after qrl_chan_sync() or behind
 * qrl_chan_stream_wait(), like the 4FSK outputs.  A handle created on the caller's hip_stream always runs the serial order.
 * (environment: QRL_CHAN_SERIAL_TAIL=1 at qrl_chan_create, for A/B runs of unmodified callers) */
enum { QRL_CHAN_OPT_LEGACY_PFB = 1, QRL_CHAN_OPT_LEGACY_TAIL = 2, QRL_CHAN_OPT_SERIAL_TAIL = 3 };
int qrl_chan_set_option(qrl_chan* c, int option, int value);
int qrl_chan_set_level(qrl_chan* c, float level);   /* _level_control multiply_const_ff, gr_demod_mmdvm_multi2.cpp:84 */
/* replaces: gr_demod_mmdvm_multi2::calibrate_rssi / gr_demod_mmdvm::calibrate_rssi -> rssi_tag_block::calibrate_rssi
 * (src/gr/gr_demod_mmdvm_multi2.cpp:138-144, src/gr/gr_demod_mmdvm.cpp:64-67) */
int qrl_chan_calibrate_rssi(qrl_chan* c, float level);
/* replaces: the RSSI stream tags of rssi_tag_block::work (src/gr/rssi_tag_block.cpp:43-68): one dB value per 300 samples at
 * 24 ksps.  rssi[(b*channel_count + ch)*cap + k], k < counts[b*channel_count + ch], holds the tags completed by each
 * following qrl_chan_process call (device pointers; NULL switches the block off). */
int qrl_chan_set_rssi_output(qrl_chan* c, float* rssi, size_t cap, uint32_t* counts);
/* optional 4FSK symbol tail behind every produced channel (BASELINE config 4: channelizer + 4FSK demod): the chain of
 * gr_demod_dmr after its resampler (src/gr/gr_demod_dmr.cpp:62-105: quadrature_demod_cf(24000/(pi/2*4800)) ->
 * fft_filter_fff(RRC(1, 24k, 4.8k, 0.2, 125)) -> symbol_sync_ff(M&M, 5 sps, 4-level) -> x0.9 -> phase_modulator -> slicer ->
 * map{3,1,2,0}) applied to the 24 ksps channel signal (after _filter[i], gr_demod_mmdvm_multi2.cpp:62-63,75).
 * bits[s*bits_cap + k]: two bits per symbol; constellation (cf32, may be NULL); counts[s*4 + 1] = symbols, [s*4 + 2] = bits
 * produced by each following qrl_chan_process call, s = b*channel_count + ch.  bits == NULL switches the tail off.
 * The symbol synchroniser runs on an internal stream of the handle (it overlaps the next call's channelizer): bits, constellation
 * and counts of a call are valid after qrl_chan_sync() or, on the device, behind qrl_chan_stream_wait() -- synchronising only the
 * cfg.hip_stream the caller passed in is NOT enough for these three outputs (it is for the int16 / RSSI outputs of such a handle). */
int qrl_chan_set_4fsk_output(qrl_chan* c, uint8_t* bits, size_t bits_cap, float* constellation, size_t constellation_cap, uint32_t* counts);
size_t qrl_chan_out_cap(const qrl_chan* c, size_t n);   /* int16 samples per channel a call with n inputs can produce */
/* replaces one scheduler pass of the multi-carrier graph: iq[b*stride + i] device cf32, n a multiple of num_channels;
 * out[(b*channel_count + c)*out_cap + k] device int16 @24 ksps, counts[b*channel_count + c] = samples written. */
int qrl_chan_process(qrl_chan* c, const float* iq, size_t stride, size_t n, int16_t* out, size_t out_cap, uint32_t* counts);
int qrl_chan_sync(qrl_chan* c);
/* ---- the two halves of qrl_chan_process for a CHANNEL-SHARDED multi-GPU job (PFB form: "channelize, then scatter the
 * channel streams"; reference: one channelizer feeds per-channel chains, src/gr/gr_demod_mmdvm_multi2.cpp:98-135, and every channel
 * has its own sink socket, src/gr/gr_mmdvm_sink.cpp:77-173).  Each rank channelizes ITS wideband inputs, an all-to-all moves every
 * channel's samples to the rank that owns the channel, and that rank runs the per-channel chains:
 *   qrl_chan_channelize        (PFB handle, form 0) stream_to_streams + pfb_channelizer_ccf only.  chan_out = device cf32,
 *                              [groups][batch][channel_count / groups][pitch]: the rows of destination rank g are contiguous, n / M
 *                              valid items per row.  The handle keeps the filter history; it must not be mixed with qrl_chan_process.
 *   qrl_chan_process_channels  (form 3 handle: `batch` = number of 25 ksps channel streams, num_channels ignored) the per-channel chain
 *                              of qrl_chan_process on chan_in[row * pitch + i], i < n1; outputs as qrl_chan_process with
 *                              channel_count = 1 (row index = stream index).  The per-channel kernel reads chan_in IN PLACE (no copy
 *                              into the handle's rings; the handle keeps the ~1.6 k samples per row it needs of it for the next call):
 *                              the buffer must stay untouched until the handle's stream has passed this call (an event recorded on
 *                              qrl_chan_stream() after the call, or qrl_chan_sync).
 *   qrl_chan_wait_for          the handle's stream waits (on the device) for what the given stream has queued so far: the collective. */
int qrl_chan_channelize(qrl_chan* c, const float* iq, size_t stride, size_t n, float* chan_out, size_t pitch, int groups);
int qrl_chan_process_channels(qrl_chan* c, const float* chan_in, size_t pitch, size_t n1, int16_t* out, size_t out_cap, uint32_t* counts);
int qrl_chan_wait_for(qrl_chan* c, void* hip_stream);
/* like qrl_demod_stream_wait: the caller's stream waits, on the device, for everything this handle has enqueued so far -- how a C4
 * caller chains its own copies / collectives (the all-to-all of a multi-GPU job) behind a call without a host synchronisation. */
int qrl_chan_stream_wait(qrl_chan* c, void* hip_stream);
void* qrl_chan_stream(qrl_chan* c);   /* hipStream_t the handle enqueues on */
int qrl_chan_internal_streams(qrl_chan* c, void* out[3]);   /* profiling aid, as qrl_demod_internal_streams: out[0] the handle's stream, out[1] the symbol-sync stream (or NULL), out[2] the per-channel kernel's stream (or NULL) */
/* like qrl_demod_profile / qrl_demod_profile_read: HIP events on the handle's stream around the kernel(s) that read the caller's
 * wideband IQ (k_pfb_chan; forms 1 / 2: the per-channel decimator launches of a call, summed) -- bench.py's roofline leg */
int qrl_chan_profile(qrl_chan* c, int enable);
int qrl_chan_profile_read(qrl_chan* c, double* kernel_ms, uint64_t* launches, const char** kernel_name);
/* the same for the three kernels of a C4 call, each timed with HIP events on the stream it is launched on: ms[0] / launches[0] the
 * kernel(s) qrl_chan_profile_read reports (channelizer), [1] the fused per-channel kernel k_chan_tail, [2] the symbol synchroniser
 * k_symsync_ff (sums over the profiled calls; a duration includes the time a kernel shares the chip with the others of the step) */
int qrl_chan_profile_read_kernels(qrl_chan* c, double ms[3], uint64_t launches[3]);

/* ---- multi-carrier MMDVM transmitter (reference src/gr/gr_mod_mmdvm_multi2.cpp:30-128) -------------------------------------
 * make_gr_mod_mmdvm_multi2(burst_timer, num_channels, channel_separation, use_tdma, sps, samp_rate, carrier_freq,
 * filter_width): per channel short_to_float(1, 32767) -> frequency_modulator_fc(2 pi 12500 / 24000) -> fft_filter_ccf ->
 * x0.8 -> rational_resampler_ccf(25, 24) -> pfb_synthesizer_ccf(10, ..., false) on ports {0,1,2,3,9,8,7} -> x(1 / num_channels)
 * -> bb gain.  in[(b*num_channels + ch)*stride + i], i < n: the int16 FM baseband gr_mmdvm_source hands out at 24 ksps
 * (src/gr/gr_mmdvm_source.cpp:180-243; its ZeroMQ framing and the tag-driven gr_zero_idle_bursts stay with the caller);
 * iq[2*(b*out_stride + k)]: the 250 ksps cf32 wideband signal, *produced (host) samples per stream for this call
 * (<= qrl_synth_out_cap(n)).  State carries across calls; asynchronous on the handle's stream. */
typedef struct qrl_synth qrl_synth;
typedef struct {
    int num_channels;        /* 1..7 (MAX_MMDVM_CHANNELS) */
    int filter_width;        /* 0 = 5000 (gr_mod_mmdvm_multi2.h:38-41) */
    int batch;               /* independent transmitters */
    size_t max_samples;      /* largest n of any call */
    void* hip_stream;
    float bb_gain;           /* gr_mod_mmdvm_multi2::set_bb_gain, 0 = 1.0 */
    int single_carrier;      /* 1 (with num_channels = 1): make_gr_mod_mmdvm (src/gr/gr_mod_mmdvm.cpp:17-64): FM -> LPF -> x0.8 -> bb
                                gain -> rational_resampler_ccf(125, 12): no synthesizer, 24 ksps -> 250 ksps */
} qrl_synth_config;
int qrl_synth_create(qrl_ctx* ctx, const qrl_synth_config* cfg, qrl_synth** out);
void qrl_synth_destroy(qrl_synth* s);
int qrl_synth_reset(qrl_synth* s);
/* gr_zero_idle_bursts (reference src/gr/gr_zero_idle_bursts.cpp:45-84; gr_mod_mmdvm_multi2.cpp:88,108, gr_mod_mmdvm.cpp:51-58):
 * the zero_samples tags gr_mmdvm_source attaches to idle slots, as absolute runs at the rate of that block's input (25 ksps behind
 * the 25/24 resampler in the multi-carrier graph, 24 ksps behind the FM modulator in the single-carrier one; item index counted from
 * the handle's creation / last reset).  Items start .. start + count - 1 of (stream, channel) become 0 + 0j in whichever following
 * qrl_synth_process calls they fall into.  host/mmdvm_wire.h zero_idle_runs() turns the source's tags into these runs. */
/* (qrl_zero_run is declared with qrl_mod_add_zero_runs above) */
int qrl_synth_add_zero_runs(qrl_synth* h, const qrl_zero_run* runs, size_t n);
int qrl_synth_set_bb_gain(qrl_synth* s, float value);
size_t qrl_synth_out_cap(const qrl_synth* s, size_t n);
int qrl_synth_process(qrl_synth* s, const int16_t* in, size_t stride, size_t n, float* iq, size_t out_stride, size_t* produced);
int qrl_synth_sync(qrl_synth* s);

/* ---- device deframer (reference src/gr/gr_deframer_bb.cpp:24-48,83-185; instances gr_demod_base.cpp:171-178) -----------
 * make_gr_deframer_bb(modem_type): 1 = 2k modes (16/24-bit sync words, 64 bits per frame), 2 = 1k modes (0xB5, 32 bits),
 * 3 = 10k modes (384 bits).  qrl_deframer_process consumes the unpacked bits of one demodulator port (bits[b*stride + i];
 * i < n, or i < counts[b*count_stride] when counts != NULL: pass the demodulator's counts + 2 or + 3 with count_stride 4)
 * and appends what gr_deframer_bb::work pushes into its mailbox -- the sync bits followed by the frame bits -- to
 * out[b*out_cap + k], k < out_counts[b] (out_cap >= 2 n + 24 never overflows).  Search state carries across calls.
 * All pointers are device pointers; asynchronous on the handle's stream (give it the demodulator's stream to chain). */
typedef struct qrl_deframer qrl_deframer;
int qrl_deframer_create(qrl_ctx* ctx, int deframer_type, int batch, void* hip_stream, qrl_deframer** out);
void qrl_deframer_destroy(qrl_deframer* d);
int qrl_deframer_reset(qrl_deframer* d);   /* gr_deframer_bb::flush (:58-65) */
int qrl_deframer_process(qrl_deframer* d, const uint8_t* bits, size_t stride, size_t n, const uint32_t* counts, size_t count_stride,
                         uint8_t* out, size_t out_cap, uint32_t* out_counts);
int qrl_deframer_sync(qrl_deframer* d);

/* ---- L1 frame synchroniser (reference gr_modem::synchronize / findSync / packBytes, src/gr_modem.cpp:1119-1282, 980-994;
 * frame types src/layer1framing.h:8-24; mode table gr_modem::toggleRxMode :203-322) ------------------------------------------
 * Consumes the unpacked bits of a demodulator port like qrl_deframer_process and emits, per stream, the frames the
 * reference would pass to gr_modem::processReceivedData as records
 *     { uint32 frame_type (FrameTypeVoice1 0xB5, FrameTypeVoice 0xED89, FrameTypeText 0x89EDAA, ...);
 *       uint32 nbytes | _modem_sync << 16  (low 16 bits: payload bytes; high 16 bits: the _modem_sync counter at the moment the frame
 *       completed, 0..39 -- what processReceivedData's voice gate of the 1k modes tests, src/gr_modem.cpp:1376-1390);
 *       nbytes payload bytes, MSB-first packed, padded to a multiple of 4 }
 * appended to out[b*out_cap ...]; out_counts[2b] = bytes written, out_counts[2b+1] = frames.  A frame that does not fit is
 * dropped (out_cap >= n/8 + qrl_framesync_frame_bytes() + 96 + 16 per frame never overflows: a frame begun in earlier calls may
 * complete in this one).  Search state, a partial frame
 * and the _modem_sync counter carry across calls.  QRL_MODEM_M17: the M17 sync words (LSF 0x55F7, stream 0xFF5D, EOT
 * 0x555D555D, :1186-1210) and 46-byte frames (:309-313); the M17 frame decoder itself stays on the host. */
typedef struct qrl_framesync qrl_framesync;
int qrl_framesync_create(qrl_ctx* ctx, int modem_type, int batch, void* hip_stream, qrl_framesync** out);
void qrl_framesync_destroy(qrl_framesync* f);
int qrl_framesync_reset(qrl_framesync* f);
int qrl_framesync_frame_bytes(const qrl_framesync* f);   /* _rx_frame_length of the mode */
int qrl_framesync_process(qrl_framesync* f, const uint8_t* bits, size_t stride, size_t n, const uint32_t* counts, size_t count_stride,
                          uint8_t* out, size_t out_cap, uint32_t* out_counts);
/* Optional: activity[b] (device, [batch]) receives with every qrl_framesync_process call the number of bits of stream b that were collected into a
 * frame while a sync was held -- non-zero exactly when gr_modem::synchronize would return data_to_process = true for the bits of this call
 * (src/gr_modem.cpp:1121-1175: "RX active", what radiocontroller.cpp:1298 polls through gr_modem::demodulate()).  NULL switches it off. */
int qrl_framesync_set_activity_output(qrl_framesync* f, uint32_t* activity);
int qrl_framesync_sync(qrl_framesync* f);

/* ---- RSSI side output (reference src/gr/rssi_block.cpp:25-50; wired src/gr/gr_demod_base.cpp:199-200: port 0 of the current
 * demodulator -> rssi_valve -> rssi_block -> probe_signal_f) -----------------------------------------------------------------
 * rssi_block(level): complex_to_mag_squared -> moving_average_ff(2000, 1, 2000) -> single_pole_iir_filter_ff(0.04) -> nlog10_ff
 * -> multiply_const_ff(10) -> add_const_ff(level).  qrl_rssi_process consumes the port-0 items of one qrl_demod_process call
 * (filtered[b*stride + i] cf32 pairs, i < counts[b*count_stride], counts == NULL: n items for every stream; device memory) and
 * writes one dB value per item to out[b*out_cap + i] (may be NULL) and the latest value of every stream to last[b] (what
 * probe_signal_f::level() returns; may be NULL).  State (2000-item window, IIR) carries across calls; results do not depend on
 * how the stream is cut into calls (the moving average restarts its running sum at every absolute multiple of 2000 items: the
 * reference's block does so at every work() call of at most max_iter = 2000 items). */
typedef struct qrl_rssi qrl_rssi;
int qrl_rssi_create(qrl_ctx* ctx, int batch, float level, void* hip_stream, qrl_rssi** out);
void qrl_rssi_destroy(qrl_rssi* r);
int qrl_rssi_reset(qrl_rssi* r);
int qrl_rssi_set_level(qrl_rssi* r, float level);   /* rssi_block::set_level (:47-50) */
int qrl_rssi_process(qrl_rssi* r, const float* filtered, size_t stride, size_t n, const uint32_t* counts, size_t count_stride,
                     float* out, size_t out_cap, float* last, uint32_t* out_counts);
int qrl_rssi_sync(qrl_rssi* r);
void* qrl_rssi_stream(qrl_rssi* r);

/* ---- spectrum side output (reference src/gr/rx_fft.cpp:44-213, instance make_rx_fft_c(32768, WIN_BLACKMAN_HARRIS)
 * src/gr/gr_demod_base.cpp:166,185 on the device-rate IQ) ---------------------------------------------------------------------
 * qrl_fft_process = rx_fft_c::work on n new samples of every stream: while enabled and nobody is behind on reading (d_push == 0)
 * the samples are multiplied by the window into the FFT buffer; when the buffer is full the next sample triggers the transform
 * (hipFFT, batched over the streams) and volk_32fc_s32f_power_spectrum_32f (10 log10 |X / N|^2), after which the block stops
 * taking samples until qrl_fft_get_fft_data has been called.  qrl_fft_get_fft_data = rx_fft_c::get_fft_data: the spectrum with its
 * halves swapped (negative frequencies first) to fft_points[b*out_stride + i] (device), *fft_size = 0 when nothing is ready.
 * wintype: gr::fft::window::win_type (0 Hamming, 1 Hann, 2 Blackman, 3 rectangular, 4 Kaiser(6.76), 5 Blackman-Harris,
 * 6 Bartlett, 7 flat top; anything else -> Hamming, :200-203). */
typedef struct qrl_fft qrl_fft;
int qrl_fft_create(qrl_ctx* ctx, int batch, unsigned fftsize, int wintype, void* hip_stream, qrl_fft** out);
void qrl_fft_destroy(qrl_fft* f);
int qrl_fft_set_enabled(qrl_fft* f, int enabled);            /* :102-107 (a new block is disabled, :58) */
int qrl_fft_set_fft_size(qrl_fft* f, unsigned fftsize);      /* :134-163 */
unsigned qrl_fft_get_fft_size(const qrl_fft* f);             /* :166-169 */
int qrl_fft_set_window_type(qrl_fft* f, int wintype);        /* :172-190 */
int qrl_fft_get_window_type(const qrl_fft* f);               /* :193-196 */
int qrl_fft_process(qrl_fft* f, const float* iq, size_t stride, size_t n);
int qrl_fft_get_fft_data(qrl_fft* f, float* fft_points, size_t out_stride, unsigned* fft_size);
int qrl_fft_sync(qrl_fft* f);
void* qrl_fft_stream(qrl_fft* f);

/* ---- filter design & tables (host side, no GPU needed): what the kernels are loaded with ----
 * replaces: gr::filter::firdes::* calls at gr_demod_2fsk.cpp:82-97, gr_demod_gmsk.cpp:80-98,
 * gr_demod_qpsk.cpp:92-103, gr_demod_base.cpp:1333-1336.  taps==NULL returns the count. */
enum { QRL_WIN_HAMMING = 0, QRL_WIN_HANN = 1, QRL_WIN_BLACKMAN = 2, QRL_WIN_RECTANGULAR = 3, QRL_WIN_BLACKMAN_HARRIS = 5 };
int qrl_firdes_low_pass(double gain, double fs, double fc, double tw, int window, float* taps);
int qrl_firdes_low_pass_2(double gain, double fs, double fc, double tw, double atten_db, int window, float* taps);
int qrl_firdes_complex_band_pass(double gain, double fs, double lo, double hi, double tw, int window, float* taps_cf32);
int qrl_firdes_root_raised_cosine(double gain, double fs, double symrate, double alpha, int ntaps, float* taps);
int qrl_table_mmse(float* t129x8);
int qrl_table_atan(float* t257);
int qrl_table_tanh(float* t256);
uint64_t qrl_phase_inc_to_turn(double radians_per_sample);

/* ---- analogue voice modulator: replaces make_gr_mod_nbfm(sps, samp_rate, carrier_freq, filter_width) (reference
 * src/gr/gr_mod_nbfm.cpp:19-77, instances gr_mod_base.cpp:171-172) for `batch` independent radios.  audio: device f32 at 8 ksps,
 * stream b at audio + b * stride, n samples per call (a multiple of 4, <= max_samples); iq: device cf32 at 1 Msps, stream b at
 * iq + 2 * b * out_stride floats, n * qrl_amod_samples_per_sample() (= 125 n) samples.  Asynchronous on the handle's stream;
 * results are independent of how the audio is cut into calls.   */
typedef struct qrl_amod qrl_amod;
typedef struct qrl_amod_config {
    int modem_type;        /* QRL_MODEM_NBFM2500 | QRL_MODEM_NBFM5000 | QRL_MODEM_USB2500 | QRL_MODEM_LSB2500 | QRL_MODEM_CW600USB | QRL_MODEM_AM5000 */
    int batch;
    size_t max_samples;    /* audio samples per stream and call */
    void* hip_stream;      /* hipStream_t or NULL (own stream) */
    float bb_gain;         /* gr_mod_nbfm::set_bb_gain; 0 = 1.0 */
    /* gr_mod_base back end (src/gr/gr_mod_base.cpp:38,215-258), as in qrl_mod_config: 0 / 1000000 = none; >= 2e6 (a multiple of 1e6): the chain's 1 Msps
     * output goes through rotator_cc(2 pi offset / 1e6) and rational_resampler_ccf(rate / 1e6, 1, low_pass(interp, rate, 480000, 20000, BH)); a non-zero
     * initial offset alone gives the rotator only.  All sample counts (samples_per_sample, out_cap, last_count) then count device-rate samples. */
    int device_samp_rate;
    double carrier_offset_hz;
} qrl_amod_config;
int qrl_amod_create(qrl_ctx* ctx, const qrl_amod_config* cfg, qrl_amod** out);
void qrl_amod_destroy(qrl_amod* m);
int qrl_amod_reset(qrl_amod* m);
int qrl_amod_set_bb_gain(qrl_amod* m, float value);
/* replaces gr_mod_base::set_carrier_offset (src/gr/gr_mod_base.cpp:799-805) for handles created with the back end; phase-continuous like rotator_cc::set_phase_inc */
int qrl_amod_set_carrier_offset(qrl_amod* m, double carrier_offset_hz);
/* replaces gr_mod_nbfm::set_ctcss(value) (src/gr/gr_mod_nbfm.cpp:101-135; gr_mod_base::set_ctcss :872-877 forwards to both NBFM instances):
 * tone_hz != 0: _audio_amplify 0.85, the audio filter becomes band_pass_2(1, 8000, 300, 3500, 200, 35, BH) and analog::sig_source_f(8000,
 * GR_COS_WAVE, tone, 0.15) is added to the audio in front of the pre-emphasis; 0: the low-pass again and _audio_amplify 0.98 (sic: the
 * constructor's 0.99 does not come back).  Takes effect with the next qrl_amod_process call; the tone's phase runs over the samples produced
 * while it is on.  NBFM handles only.  [The tone source is GNU Radio's fixed-point NCO with its 1024-row sine table, restated from memory.] */
int qrl_amod_set_ctcss(qrl_amod* m, float tone_hz);
/* replaces gr_mod_base::set_filter_width(filter_width, mode) (src/gr/gr_mod_base.cpp:878-905) -> gr_mod_nbfm::set_filter_width (gr_mod_nbfm.cpp:78-93:
 * _if_resampler low_pass_2(25, 200000, w, w, 60, BH), _filter low_pass_2(1, 50000, w, 1200, 60, BH), _resampler low_pass_2(sps, 1e6, w, w, 60, BH),
 * sensitivity 4 pi w / 50000 -- the constructor uses a 3500 Hz transition everywhere), gr_mod_am (gr_mod_am.cpp:75-85: the constructor's designs with w),
 * gr_mod_ssb (gr_mod_ssb.cpp:85-100: _resampler as constructed with w, the sideband filter complex_band_pass_2(1, 8000, 300, w | -w, -300, 250, 90, BH);
 * the audio filter keeps the constructor's width).  The chain restarts from a fresh state (see qrl_demod_set_filter_width); bb_gain and the CTCSS
 * switch are kept.  Widths whose filters do not fit the kernels' tables are refused (NBFM < 540, SSB < 500, AM < 295 Hz). */
int qrl_amod_set_filter_width(qrl_amod* m, int filter_width);
/* QRL_MODEM_CW600USB (TX only): replaces the CW branch of gr_mod_base (src/gr/gr_mod_base.cpp:144,180,679-683): _signal_source = analog::sig_source_f(8000,
 * GR_SIN_WAVE, 600, 0.001, 1) -> _usb_cw = make_gr_mod_ssb(125, 1000000, 1700, 1000, 0).  qrl_amod_process(m, NULL, 0, n, iq, stride) produces what n samples
 * of the tone source give (whole chunks of 1024 like every SSB handle; an audio pointer is ignored); qrl_amod_set_cw_k replaces gr_mod_base::set_cw_k (:948-956):
 * key down = amplitude 0.98, up = 0.001, from the next call on, the tone's phase runs on.  qrl_amod_set_filter_width works as for USB.  [The tone source is
 * GNU Radio's fixed-point NCO with its 1024-row sine table, restated from memory.] */
int qrl_amod_set_cw_k(qrl_amod* m, int key_down);
size_t qrl_amod_samples_per_sample(const qrl_amod* m);
/* SSB (replaces make_gr_mod_ssb(125, 1000000, 1700, 2700, sb), reference src/gr/gr_mod_ssb.cpp:19-82, gr_mod_base.cpp:178-179): the
 * cessb stretcher emits whole chunks of 1024 audio-rate items and looks two items ahead, so a call returns 125 x (chunks completed
 * by it) samples -- any n is accepted; qrl_amod_last_count = IQ samples per stream the last call wrote, qrl_amod_out_cap(n) = the
 * bound for a call of n audio samples (what out_stride must hold).  NBFM: both are 125 n. */
size_t qrl_amod_last_count(const qrl_amod* m);
size_t qrl_amod_out_cap(const qrl_amod* m, size_t n);
int qrl_amod_process(qrl_amod* m, const float* audio, size_t stride, size_t n, float* iq, size_t out_stride);
int qrl_amod_sync(qrl_amod* m);
void* qrl_amod_stream(qrl_amod* m);

/* ---- frame FEC of the DMR / M17 protocol stacks over batches of frames.  Stateless; device pointers;
 * asynchronous on hip_stream (NULL = the default stream), the caller synchronises it.
 * qrl_bptc19696_decode replaces CBPTC19696::decode (reference src/MMDVM/BPTC19696.cpp:47-64; users: CDMRFullLC, CDMRCSBK, CDMRDataHeader
 * in src/MMDVM): the 196 code bits of each 33-byte DMR burst -> de-interleave -> up to 5 rounds of Hamming (13,9,3) column and
 * (15,11,3) row correction -> the 96 payload bits as 12 bytes.  qrl_bptc19696_encode replaces CBPTC19696::encode (:67-87): payload
 * -> row / column parities -> interleave -> written into the code-bit positions of the burst (its other bits are kept). */
int qrl_bptc19696_decode(qrl_ctx* ctx, void* hip_stream, const uint8_t* bursts /* [n][33] */, size_t n, uint8_t* payloads /* [n][12] */);
int qrl_bptc19696_encode(qrl_ctx* ctx, void* hip_stream, const uint8_t* payloads /* [n][12] */, size_t n, uint8_t* bursts /* [n][33] in/out */);
/* qrl_m17_decode_frames replaces the per-frame work of M17FrameDecoder::decodeFrame (reference src/M17/M17/M17FrameDecoder.cpp:44-215:
 * decorrelate, de-interleave, sync-word classification, punctured K = 5 Viterbi of the LSF / stream payload, Golay(24,12) of the LICH).
 * frames: [n][48] (sync word + 46 bytes, as the frame synchroniser delivers them).  records: [n][QRL_M17_RECORD_BYTES] =
 * {type (M17FrameType: 0 preamble, 1 link setup, 2 stream, 4 unknown), LICH ok, payload[30] (30 LSF bytes | 18 stream-frame bytes),
 * LICH segment[6] (5 LSF bytes + segment number), 0, 0}.  The LSF reassembly from LICH segments (:130-147, per-stream state + CRC)
 * is host work: host/m17_frame_decoder_hip. */
#define QRL_M17_RECORD_BYTES 40
int qrl_m17_decode_frames(qrl_ctx* ctx, void* hip_stream, const uint8_t* frames, size_t n, uint8_t* records);
/* the inverse: replaces the per-frame work of M17FrameEncoder::encodeLsf / encodeStreamFrame (reference
 * src/M17/M17/M17FrameEncoder.cpp:52-118): records of the same layout (type 1: the 30 LSF bytes with their CRC; type 2: the 18
 * stream-frame bytes = frame number (EOS bit in its top bit) + 16 payload bytes, and the 6-byte LICH segment = 5 LSF bytes +
 * segment number) -> 48-byte frames.  Frame numbering, the LICH round robin and the LSF CRC are per-stream state: host work. */
int qrl_m17_encode_frames(qrl_ctx* ctx, void* hip_stream, const uint8_t* records, size_t n, uint8_t* frames);

/* ---- developer aids (NOT part of the drop-in surface; tools/prof_phases.py): shader-clock phase profile of k_decim_mfma ---- */
void qrl_debug_decim_prof(unsigned long long* out8);
void qrl_debug_decim_prof_enable(int on);

#ifdef __cplusplus
}
#endif
#endif
