// amod.cpp — analogue voice modulators behind the C ABI: gr_mod_nbfm, and gr_mod_ssb (reference src/gr/gr_mod_ssb.cpp:26-82, instances
// gr_mod_base.cpp:178-179: audio filter -> float_to_complex -> cessb clipper -> stretcher -> side-band filter -> gains -> 1:125).
// gr_mod_nbfm (reference src/gr/gr_mod_nbfm.cpp:26-77, instances
// make_gr_mod_nbfm(20, 1000000, 1700, 2500 / 5000) src/gr/gr_mod_base.cpp:171-172) for a batch of independent radios.
//   audio (8 ksps) -> audio filter -> x0.99 -> pre-emphasis (iir_filter_ffd, f64) -> 25:4 -> frequency modulator -> channel filter
//   -> x0.8 -> x bb_gain -> 1:20 interpolator  = 125 IQ samples (1 Msps) per audio sample.
// Kernels: k_am_load, k_fir_fff, k_am_iir, k_an_resamp (host-side counts), k_tx_fm, k_fir_ccf, k_scale_c, k_tx_interp_c.
// Arithmetic = oracle/orc_chains.c orc_mod_nbfm, bit for bit, independent of how the audio is cut into calls (multiples of 4 samples:
// the 25:4 resampler then produces whole groups of 25).
#include "../../include/qrl_hip.h"
#include "engine.hpp"
#include "firdes.hpp"
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

using namespace qrl;
extern int qrl_set_error(int code, const std::string& msg);
struct qrl_ctx { int device; };

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return qrl_set_error(QRL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

namespace {
template <class T> struct Dev {
    T* p = nullptr; size_t n = 0;
    ~Dev() { if (p) (void)hipFree(p); }
    int alloc(size_t count) { if (p) { (void)hipFree(p); p = nullptr; } n = count; return hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T)) == hipSuccess ? QRL_OK : QRL_ERR_HIP; }
    int upload(const std::vector<T>& v) { if (int r = alloc(v.size())) return r; return hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) == hipSuccess ? QRL_OK : QRL_ERR_HIP; }
    int zero() { return hipMemset(p, 0, n * sizeof(T)) == hipSuccess ? QRL_OK : QRL_ERR_HIP; }
};
uint32_t pow2_at_least(size_t v) { uint32_t c = 1024; while (c < v) c <<= 1; return c; }
// gr::fxpt's 1024-row interpolated sine table {slope, intercept} (oracle orc_fxpt_sine_table)
std::vector<float> fxpt_sine_table()
{
    std::vector<float> tab(2048);
    for (int i = 0; i < 1024; ++i) {
        const double a = (double)i * 2097152.0, b = (double)(i + 1) * 2097152.0, w = M_PI / 1073741824.0;
        const double fa = std::sin(a * w), fb = std::sin(b * w), fm = std::sin((a + b) / 2 * w);
        tab[2 * i] = (float)((fb - fa) / (b - a));
        tab[2 * i + 1] = (float)((3 * a + b) * (fa - fb) / (4 * (b - a)) + (fm + fa) / 2);
    }
    return tab;
}
// sig_source_f::set_frequency -> fxpt_nco::set_freq((float)(2 pi f / fs)) -> float_to_fixed (oracle orc_fxpt_phase_inc)
uint32_t fxpt_phase_inc(double fs, double freq)
{
    float x = (float)(2 * M_PI * freq / fs);
    const float PI_F = (float)M_PI;
    const int d = (int)std::floor(x / 2 / PI_F + 0.5f);
    x -= d * 2 * PI_F;
    return (uint32_t)(int32_t)(x * 2147483648.0f / PI_F);
}
}  // namespace

struct qrl_amod {
    qrl_ctx* ctx = nullptr;
    qrl_amod_config cfg{};
    hipStream_t stream = nullptr; bool own_stream = false;
    int sps = 20, fw = 5000; bool ssb = false, lsb = false, am = false;
    bool cw = false; double cw_ampl = 0.001;   // QRL_MODEM_CW600USB: the SSB chain fed by sig_source_f(8000, GR_SIN_WAVE, 600, 0.001, 1) (gr_mod_base.cpp:144,180,679-683)
    Dev<float> am_gain; Dev<float2> t_chan, m1, m2; int n_chan = 0; uint32_t mm = 0; uint64_t n1m = 0;   // AM: agc gain per stream, 1 Msps rings, channel filter
    Dev<float2> t_side, c1, c2, c3; int n_side = 0; Dev<float> atan_tab; uint64_t ns = 0; size_t last = 0;   // SSB
    float bb_gain = 1.0f, fm_k = 0.f;
    Dev<float> t_audio, t_if, t_filt, t_interp; int n_audio = 0, n_if = 0, n_filt = 0, n_interp = 0;
    // gr_mod_nbfm::set_ctcss (src/gr/gr_mod_nbfm.cpp:101-135): band-pass audio filter, _audio_amplify 0.85 / 0.98, tone source + add_ff
    Dev<float> t_audio_bp, tone_tab; int n_audio_bp = 0; float k_audio = 0.99f; float tone_hz = 0.0f; uint32_t tone_inc = 0; uint64_t tone_k = 0, cw_k = 0;   // sample counters of the CTCSS tone and of the CW key's tone source (sig_source_f free-runs: a filter change does not reset them)
    Dev<float> a0, a1, a2, r50; uint32_t m8 = 0, m50 = 0;       // rings: audio in, filtered, pre-emphasised (8 ksps); 50 ksps
    Dev<float2> fmv, flt;                                        // 50 ksps complex: modulator out, channel filter out
    Dev<AmIirState> iir; Dev<float> phase;
    double pb[2] = {0, 0}, pa[2] = {0, 0};
    uint64_t n8 = 0, n50 = 0;
    // gr_mod_base back end (src/gr/gr_mod_base.cpp:38,215-258), as behind the digital modulators (tx.cpp): rotator at 1 Msps, then the interpolator to the device rate
    bool backend = false; int be_interp = 1, be_nt = 0; Dev<float> be_taps;
    Dev<float2> bb, be_ring, rot_lo; size_t bb_stride = 0; uint32_t be_mask = 0;
    uint64_t rot_inc = 0, rot_acc = 0, rot_nbase = 0, n_bb = 0;
    int set_rot(double hz)
    {
        rot_inc = phase_inc_to_turn(2 * M_PI * hz / 1000000.0);
        std::vector<float2> lo(512);
        for (int r = 0; r < 512; ++r) { float sn, cs; sincos_turn_host((uint64_t)r * rot_inc, sn, cs); lo[r] = make_float2(cs, sn); }
        return hipMemcpy(rot_lo.p, lo.data(), 512 * sizeof(float2), hipMemcpyHostToDevice) == hipSuccess ? QRL_OK : QRL_ERR_HIP;
    }
    size_t cap_1msps(size_t n) const { return am ? n * (size_t)sps : ssb ? (n + 1024) * (size_t)sps : n * 25 / 4 * (size_t)sps; }
    int init_back_end()
    {
        const int rate = cfg.device_samp_rate;
        if (rate != 0 && rate != 1000000 && (rate < 2000000 || rate % 1000000 != 0 || rate > 64000000))
            return qrl_set_error(QRL_ERR_ARG, "amod: device_samp_rate must be 1e6 or a multiple of 1e6 in [2e6, 64e6]");
        be_interp = rate >= 2000000 ? rate / 1000000 : 1;
        backend = be_interp > 1 || cfg.carrier_offset_hz != 0.0;
        if (!backend) return QRL_OK;
        int r;
        bb_stride = cap_1msps(cfg.max_samples);
        if ((r = bb.alloc((size_t)cfg.batch * bb_stride)) || (r = rot_lo.alloc(512)) || (r = set_rot(cfg.carrier_offset_hz))) return r;
        if (be_interp > 1) {
            const std::vector<float> lp = low_pass(be_interp, rate, 480000, 20000, WIN_BLACKMAN_HARRIS);   // gr_mod_base.cpp:215-258
            be_nt = (int)lp.size();
            if ((r = be_taps.upload(lp))) return r;
            be_mask = pow2_at_least(bb_stride + (size_t)be_nt / be_interp + 64) - 1;
            if ((r = be_ring.alloc((size_t)cfg.batch * (be_mask + 1)))) return r;
        }
        return QRL_OK;
    }
    ~qrl_amod() { if (own_stream && stream) (void)hipStreamDestroy(stream); }
    int init_state(bool keep_tone_phase = false)
    {
        int r;
        for (auto* b : {&a0, &a1, &a2, &r50, &phase}) if ((r = b->zero())) return r;
        if ((r = fmv.zero()) || (r = flt.zero()) || (r = iir.zero())) return r;
        if (c1.p && ((r = c1.zero()) || (r = c2.zero()) || (r = c3.zero()))) return r;
        if (am) {
            if ((r = m1.zero()) || (r = m2.zero())) return r;
            std::vector<float> one((size_t)cfg.batch, 1.0f);                      // agc2_ff(..., gain = 1)
            if (hipMemcpy(am_gain.p, one.data(), one.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return QRL_ERR_HIP;
            n1m = 0;
        }
        n8 = n50 = ns = 0; last = 0;
        if (!keep_tone_phase) tone_k = cw_k = 0;   // qrl_amod_set_filter_width restarts the chain but the reference's sig_source_f keeps running
        if (be_ring.p && (r = be_ring.zero())) return r;
        n_bb = 0; rot_acc = 0; rot_nbase = 0;
        return QRL_OK;
    }
};

extern "C" {

int qrl_amod_create(qrl_ctx* ctx, const qrl_amod_config* cfg, qrl_amod** outp)
{
    if (!ctx || !cfg || !outp) return QRL_ERR_ARG;
    if (cfg->batch < 1 || cfg->max_samples < 4) return qrl_set_error(QRL_ERR_ARG, "amod: batch >= 1, max_samples >= 4");
    std::unique_ptr<qrl_amod> m(new (std::nothrow) qrl_amod);
    if (!m) return QRL_ERR_NOMEM;
    m->ctx = ctx; m->cfg = *cfg; m->bb_gain = cfg->bb_gain == 0.0f ? 1.0f : cfg->bb_gain;
    switch (cfg->modem_type) {
    case QRL_MODEM_NBFM2500: m->fw = 2500; break;     // make_gr_mod_nbfm(20, 1000000, 1700, 2500) gr_mod_base.cpp:171
    case QRL_MODEM_NBFM5000: m->fw = 5000; break;     // :172
    case QRL_MODEM_USB2500: m->ssb = true; m->fw = 2700; m->sps = 125; break;              // make_gr_mod_ssb(125, 1000000, 1700, 2700, 0) :178
    case QRL_MODEM_LSB2500: m->ssb = m->lsb = true; m->fw = 2700; m->sps = 125; break;     // :179
    case QRL_MODEM_AM5000: m->am = true; m->fw = 5000; m->sps = 125; break;                 // make_gr_mod_am(125, 1000000, 1700, 5000) gr_mod_base.cpp:167
    case QRL_MODEM_CW600USB: m->ssb = m->cw = true; m->fw = 1000; m->sps = 125; break;     // _usb_cw = make_gr_mod_ssb(125, 1000000, 1700, 1000, 0) :180
    default: return qrl_set_error(QRL_ERR_ARG, "amod: modem_type must be QRL_MODEM_NBFM2500 / NBFM5000 / USB2500 / LSB2500 / CW600USB / AM5000");
    }
    HIPCHK(hipSetDevice(ctx->device));
    if (cfg->hip_stream) m->stream = static_cast<hipStream_t>(cfg->hip_stream);
    else { HIPCHK(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking)); m->own_stream = true; }
    const int fw = m->fw, B = cfg->batch;
    if (m->am) {
        // gr_mod_am.cpp:40-57: audio band-pass, interpolator low_pass(sps, samp_rate, fw, fw) (Hamming), channel filter complex_band_pass_2
        const std::vector<float> ta = band_pass_2(1, 8000, 300, 3000, 200, 60, WIN_HAMMING);
        const std::vector<float> tr = low_pass(m->sps, 1000000, fw, fw, WIN_HAMMING);
        const auto tc = complex_band_pass_2(1, 1000000, -fw, fw, 1200, 120, WIN_BLACKMAN_HARRIS);
        m->n_audio = (int)ta.size(); m->n_interp = (int)tr.size(); m->n_chan = (int)tc.size();
        if (m->n_interp > 8192 || (size_t)m->n_chan * sizeof(float2) > 60 * 1024) return qrl_set_error(QRL_ERR_ARG, "amod: AM filters too long for the kernels' tables");
        std::vector<float2> tc2(tc.size());
        for (size_t i = 0; i < tc.size(); ++i) tc2[i] = make_float2(tc[i].real(), tc[i].imag());
        int r;
        if ((r = m->t_audio.upload(ta)) || (r = m->t_interp.upload(tr)) || (r = m->t_chan.upload(tc2))) return r;
        m->m8 = pow2_at_least(cfg->max_samples + 1024) - 1;
        m->mm = pow2_at_least((size_t)cfg->max_samples * m->sps + (size_t)m->n_chan + 1024) - 1;
        const size_t r8 = (size_t)B * (m->m8 + 1), r1m = (size_t)B * (m->mm + 1);
        if ((r = m->a0.alloc(r8)) || (r = m->a1.alloc(r8)) || (r = m->a2.alloc(r8)) || (r = m->c1.alloc(r8)) || (r = m->c2.alloc(1)) || (r = m->c3.alloc(1)) ||
            (r = m->m1.alloc(r1m)) || (r = m->m2.alloc(r1m)) || (r = m->am_gain.alloc(B))) return r;
        if ((r = m->r50.alloc(1)) || (r = m->fmv.alloc(1)) || (r = m->flt.alloc(1)) || (r = m->iir.alloc(1)) || (r = m->phase.alloc(1))) return r;
        if ((r = m->init_back_end()) || (r = m->init_state())) return r;
        *outp = m.release();
        return QRL_OK;
    }
    if (m->ssb) {
        const std::vector<float> ta = band_pass_2(1, 8000, 300, fw, 200, 90, WIN_BLACKMAN_HARRIS);             // _audio_filter, gr_mod_ssb.cpp:43-45
        const auto ts = m->lsb ? complex_band_pass_2(1, 8000, -fw, -200, 200, 90, WIN_BLACKMAN_HARRIS)           // _filter_lsb, :56-57
                               : complex_band_pass_2(1, 8000, 200, fw, 200, 90, WIN_BLACKMAN_HARRIS);            // _filter_usb, :54-55
        const std::vector<float> tr = low_pass_2(m->sps, 1000000, fw, fw, 90, WIN_BLACKMAN_HARRIS);            // _resampler (125, 1), :47-50
        m->n_audio = (int)ta.size(); m->n_side = (int)ts.size(); m->n_interp = (int)tr.size();
        if (m->n_interp > 8192) return qrl_set_error(QRL_ERR_ARG, "amod: interpolator filter too long");   // (beyond 2048 taps k_tx_interp_c reads them through L1 / L2: the 4091 taps of the CW chain)
        std::vector<float2> ts2(ts.size());
        for (size_t i = 0; i < ts.size(); ++i) ts2[i] = make_float2(ts[i].real(), ts[i].imag());
        int r;
        if ((r = m->t_audio.upload(ta)) || (r = m->t_side.upload(ts2)) || (r = m->t_interp.upload(tr)) || (r = m->atan_tab.upload(atan_table()))) return r;
        if (m->cw) { if ((r = m->tone_tab.upload(fxpt_sine_table()))) return r; m->tone_inc = fxpt_phase_inc(8000.0, 600.0); }
        m->m8 = pow2_at_least(cfg->max_samples + 2048) - 1;        // the stretcher holds back up to 1025 items
        const size_t r8 = (size_t)B * (m->m8 + 1);
        if ((r = m->a0.alloc(r8)) || (r = m->a1.alloc(r8)) || (r = m->c1.alloc(r8)) || (r = m->c2.alloc(r8)) || (r = m->c3.alloc(r8))) return r;
        // (members of the FM chain stay empty)
        if ((r = m->a2.alloc(1)) || (r = m->r50.alloc(1)) || (r = m->fmv.alloc(1)) || (r = m->flt.alloc(1)) || (r = m->iir.alloc(1)) || (r = m->phase.alloc(1))) return r;
        if ((r = m->init_back_end()) || (r = m->init_state())) return r;
        *outp = m.release();
        return QRL_OK;
    }
    const std::vector<float> ta = low_pass_2(1, 8000, 3500, 200, 35, WIN_BLACKMAN_HARRIS);                 // _audio_filter, gr_mod_nbfm.cpp:43-45
    const std::vector<float> ti = low_pass_2(25, 50000.0 * 4, fw, 3500, 60, WIN_BLACKMAN_HARRIS);          // _if_resampler (25, 4), :49-51
    const std::vector<float> tf = low_pass_2(1, 50000, fw, 3500, 60, WIN_BLACKMAN_HARRIS);                 // _filter, :61-62
    const std::vector<float> tr = low_pass_2(m->sps, 1000000, fw, 3500, 60, WIN_BLACKMAN_HARRIS);          // _resampler (sps, 1), :56-58
    m->n_audio = (int)ta.size(); m->n_if = (int)ti.size(); m->n_filt = (int)tf.size(); m->n_interp = (int)tr.size();
    if (m->n_interp > 8192) return qrl_set_error(QRL_ERR_ARG, "amod: interpolator filter too long");   // (beyond 2048 taps k_tx_interp_c reads them through L1 / L2: the 4091 taps of the CW chain)
    int r;
    if ((r = m->t_audio.upload(ta)) || (r = m->t_if.upload(ti)) || (r = m->t_filt.upload(tf)) || (r = m->t_interp.upload(tr))) return r;
    {   // set_ctcss(tone): the band-pass and the sine table of the tone source (oracle orc_fxpt_sine_table), ready for qrl_amod_set_ctcss
        const std::vector<float> tb = band_pass_2(1, 8000, 300, 3500, 200, 35, WIN_BLACKMAN_HARRIS);        // gr_mod_nbfm.cpp:124-125
        m->n_audio_bp = (int)tb.size();
        if ((r = m->t_audio_bp.upload(tb)) || (r = m->tone_tab.upload(fxpt_sine_table()))) return r;
    }
    m->fm_k = (float)(4 * M_PI * fw / 50000.0f);                                                          // frequency_modulator_fc, :41
    preemph_taps(8000, 50e-6, m->pa, m->pb);                                                              // :39
    const size_t max50 = cfg->max_samples * 25 / 4 + 32;
    m->m8 = pow2_at_least(cfg->max_samples + 1024) - 1;
    m->m50 = pow2_at_least(max50 + 2048) - 1;
    const size_t r8 = (size_t)B * (m->m8 + 1), r5 = (size_t)B * (m->m50 + 1);
    if ((r = m->a0.alloc(r8)) || (r = m->a1.alloc(r8)) || (r = m->a2.alloc(r8)) || (r = m->r50.alloc(r5)) || (r = m->fmv.alloc(r5)) ||
        (r = m->flt.alloc(r5)) || (r = m->iir.alloc(B)) || (r = m->phase.alloc(B))) return r;
    if ((r = m->init_back_end()) || (r = m->init_state())) return r;
    *outp = m.release();
    return QRL_OK;
}
void qrl_amod_destroy(qrl_amod* m) { if (m) { (void)hipStreamSynchronize(m->stream); delete m; } }
int qrl_amod_reset(qrl_amod* m)
{
    if (!m) return QRL_ERR_ARG;
    HIPCHK(hipStreamSynchronize(m->stream));
    return m->init_state();
}
int qrl_amod_set_bb_gain(qrl_amod* m, float g) { if (!m) return QRL_ERR_ARG; m->bb_gain = g; return QRL_OK; }
int qrl_amod_set_ctcss(qrl_amod* m, float tone_hz)
{
    if (!m) return QRL_ERR_ARG;
    if (m->ssb || m->am) return qrl_set_error(QRL_ERR_ARG, "qrl_amod_set_ctcss: NBFM modulators only (gr_mod_base::set_ctcss forwards to the two gr_mod_nbfm instances)");
    if (tone_hz < 0.0f || tone_hz > 300.0f) return qrl_set_error(QRL_ERR_ARG, "qrl_amod_set_ctcss: tone out of range");
    if (tone_hz == 0.0f) { m->k_audio = 0.98f; m->tone_hz = 0.0f; return QRL_OK; }      // gr_mod_nbfm.cpp:104-108 (0.98, not the constructor's 0.99)
    m->k_audio = 0.85f; m->tone_hz = tone_hz;                                             // :122-126
    m->tone_inc = fxpt_phase_inc(8000.0, (double)tone_hz);
    return QRL_OK;
}
int qrl_amod_set_cw_k(qrl_amod* m, int key_down)
{
    if (!m || !m->cw) return qrl_set_error(QRL_ERR_ARG, "qrl_amod_set_cw_k: QRL_MODEM_CW600USB handles only");
    m->cw_ampl = key_down ? 0.98 : 0.001;   // gr_mod_base::set_cw_k -> sig_source_f::set_amplitude (gr_mod_base.cpp:948-956): from the next call on, the phase runs on
    return QRL_OK;
}
int qrl_amod_set_filter_width(qrl_amod* m, int width)
{
    if (!m) return QRL_ERR_ARG;
    // the setters' own designs (they do not repeat the constructors'): gr_mod_nbfm.cpp:78-93, gr_mod_am.cpp:75-85, gr_mod_ssb.cpp:85-100
    const double w = width;
    if (width <= 0 || (m->ssb ? (width <= 300 || width > 4000) : m->am ? 2 * width > 1000000 : width > 25000))
        return qrl_set_error(QRL_ERR_ARG, "qrl_amod_set_filter_width: width out of range");
    HIPCHK(hipSetDevice(m->ctx->device));
    HIPCHK(hipStreamSynchronize(m->stream));
    auto to2 = [](const std::vector<std::complex<float>>& t) { std::vector<float2> o(t.size()); for (size_t i = 0; i < t.size(); ++i) o[i] = make_float2(t[i].real(), t[i].imag()); return o; };
    int r;
    if (m->am) {
        const std::vector<float> tr = low_pass(m->sps, 1000000, w, w, WIN_HAMMING);
        const auto tc = complex_band_pass_2(1, 1000000, -w, w, 1200, 120, WIN_BLACKMAN_HARRIS);
        if (tr.size() > 8192 || tc.size() != (size_t)m->n_chan) return qrl_set_error(QRL_ERR_ARG, "qrl_amod_set_filter_width: AM interpolator too long (width >= 295)");
        if ((r = m->t_interp.upload(tr)) || (r = m->t_chan.upload(to2(tc)))) return r;
        m->n_interp = (int)tr.size();
    } else if (m->ssb) {
        // (the audio filter keeps the constructor's width)
        const std::vector<float> tr = low_pass_2(m->sps, 1000000, w, w, 90, WIN_BLACKMAN_HARRIS);
        const auto ts = m->lsb ? complex_band_pass_2(1, 8000, -w, -300, 250, 90, WIN_BLACKMAN_HARRIS) : complex_band_pass_2(1, 8000, 300, w, 250, 90, WIN_BLACKMAN_HARRIS);
        if (tr.size() > 8192 || ts.size() > 1024) return qrl_set_error(QRL_ERR_ARG, "qrl_amod_set_filter_width: filter too long (width >= 500)");
        if ((r = m->t_interp.upload(tr)) || (r = m->t_side.upload(to2(ts)))) return r;
        m->n_interp = (int)tr.size(); m->n_side = (int)ts.size();
    } else {
        const std::vector<float> ti = low_pass_2(25, 50000.0 * 4, w, w, 60, WIN_BLACKMAN_HARRIS);
        const std::vector<float> tf = low_pass_2(1, 50000, w, 1200, 60, WIN_BLACKMAN_HARRIS);
        const std::vector<float> tr = low_pass_2(m->sps, 1000000, w, w, 60, WIN_BLACKMAN_HARRIS);
        if (tr.size() > 8192 || ti.size() > 1024 || tf.size() > 1024) return qrl_set_error(QRL_ERR_ARG, "qrl_amod_set_filter_width: filter too long (width >= 540)");
        if ((r = m->t_if.upload(ti)) || (r = m->t_filt.upload(tf)) || (r = m->t_interp.upload(tr))) return r;
        m->n_if = (int)ti.size(); m->n_filt = (int)tf.size(); m->n_interp = (int)tr.size();
        m->fm_k = (float)(4 * M_PI * width / 50000.0f);
    }
    m->fw = width;
    // The reference swaps the taps of a running graph under lock() / unlock(), at a sample position its scheduler decides; here the chain restarts
    // from a fresh state (like qrl_amod_reset; the set_ctcss switch, bb_gain and the PHASE of the tone sources are kept) -- what the tests compare is the chain
    // built with the setter's designs.
    return m->init_state(true);
}
size_t qrl_amod_samples_per_sample(const qrl_amod* m) { return m ? ((m->ssb || m->am) ? (size_t)m->sps : (size_t)25 * m->sps / 4) * (size_t)m->be_interp : 0; }
size_t qrl_amod_last_count(const qrl_amod* m) { return m ? m->last : 0; }
size_t qrl_amod_out_cap(const qrl_amod* m, size_t n) { return m ? m->cap_1msps(n) * (size_t)m->be_interp : 0; }
int qrl_amod_set_carrier_offset(qrl_amod* m, double hz)
{
    if (!m) return QRL_ERR_ARG;
    if (!m->backend) return qrl_set_error(QRL_ERR_ARG, "analogue modulator was created without the gr_mod_base back end");
    HIPCHK(hipStreamSynchronize(m->stream));   // rot_lo is rewritten below
    m->rot_acc += (m->n_bb - m->rot_nbase) * m->rot_inc;   // phase-continuous, like rotator_cc::set_phase_inc
    m->rot_nbase = m->n_bb;
    return m->set_rot(hz);
}
void* qrl_amod_stream(qrl_amod* m) { return m ? m->stream : nullptr; }
int qrl_amod_sync(qrl_amod* m) { if (!m) return QRL_ERR_ARG; HIPCHK(hipStreamSynchronize(m->stream)); return QRL_OK; }

int qrl_amod_process(qrl_amod* m, const float* audio, size_t stride, size_t n, float* iq, size_t out_stride)
{
    if (!m || (!audio && n && !m->cw) || (!iq && n)) return QRL_ERR_ARG;
    if (n > m->cfg.max_samples) return qrl_set_error(QRL_ERR_TOO_BIG, "n exceeds max_samples");
    if (!m->ssb && !m->am && n % 4) return qrl_set_error(QRL_ERR_ARG, "amod: audio samples per call must be a multiple of 4 (25:4 resampler)");
    m->last = 0;
    if (n == 0) return QRL_OK;
    HIPCHK(hipSetDevice(m->ctx->device));
    const int B = m->cfg.batch;
    hipStream_t s = m->stream;
    // with the back end the chain's 1 Msps output goes to the handle's own linear buffer, and from there through the rotator (and the interpolator)
    float2* const mod_out = m->backend ? m->bb.p : reinterpret_cast<float2*>(iq);
    const size_t mod_stride = m->backend ? m->bb_stride : out_stride;
    auto back_end = [&](uint32_t n1) {   // n1 samples per stream at 1 Msps are in bb
        if (!m->backend || !n1) return;
        TxRotParams rp{}; rp.in = m->bb.p; rp.in_stride = m->bb_stride; rp.n0 = m->n_bb; rp.count = n1;
        rp.rot_acc = m->rot_acc; rp.rot_inc = m->rot_inc; rp.rot_nbase = m->rot_nbase; rp.rot_lo = m->rot_lo.p;
        if (m->be_interp > 1) rp.out_ring = RingC{m->be_ring.p, m->be_mask};
        else { rp.out = reinterpret_cast<float2*>(iq); rp.out_stride = out_stride; }
        launch_tx_rot(rp, B, s);
        if (m->be_interp > 1) {
            TxInterpCParams bp{}; bp.in = rp.out_ring; bp.n0 = m->n_bb * (uint64_t)m->be_interp; bp.count = n1 * (uint32_t)m->be_interp;
            bp.taps = m->be_taps.p; bp.nt = m->be_nt; bp.interp = m->be_interp;
            bp.out = reinterpret_cast<float2*>(iq); bp.out_stride = out_stride;
            launch_tx_interp_c(bp, B, s);
        }
        m->n_bb += n1;
    };
    if (m->am) {   // gr_mod_am.cpp:66-77 in connection order
        RingF a0{m->a0.p, m->m8}, a1{m->a1.p, m->m8}, a2{m->a2.p, m->m8};
        RingC c1{m->c1.p, m->m8}, m1{m->m1.p, m->mm}, m2{m->m2.p, m->mm};
        const uint32_t c8 = (uint32_t)n, c1m = (uint32_t)(n * (size_t)m->sps);
        // (every batch size: out_stride is the capacity of the final filter's output port, a smaller one would silently truncate -- ADVICE r4)
        if ((size_t)c1m * m->be_interp > out_stride) return qrl_set_error(QRL_ERR_ARG, "amod: out_stride smaller than this call's output (qrl_amod_out_cap)");
        AmLoadParams lp{}; lp.in = audio; lp.in_stride = stride; lp.out = a0; lp.n0 = m->n8; lp.count = c8;
        launch_am_load(lp, B, s);
        AmAgcParams ap{}; ap.in = a0; ap.out = a1; ap.n0 = m->n8; ap.count = c8; ap.attack = 1e-2f; ap.decay = 1e-4f; ap.ref = 1.0f; ap.max_gain = 1.0f;
        ap.lo = -0.98f; ap.hi = 0.98f; ap.scale = 0.95f; ap.gain = m->am_gain.p;
        launch_am_agc_rail(ap, B, s);                                               // _agc, _rail, _audio_amplify
        FirFffParams af{}; af.in = a1; af.out = a2; af.q0 = m->n8; af.count = c8; af.taps = m->t_audio.p; af.nt = m->n_audio;
        launch_fir_fff(af, B, s);                                                   // _audio_filter
        launch_am_carrier(a2, c1, m->n8, c8, 0.5f, B, s);                           // _add with _signal_source (frequency 0: a constant), _float_to_complex
        TxInterpCParams xp{}; xp.in = c1; xp.n0 = m->n1m; xp.count = c1m; xp.taps = m->t_interp.p; xp.nt = m->n_interp; xp.interp = m->sps;
        xp.out = nullptr; xp.out_stride = 0; xp.out_ring = m1;
        launch_tx_interp_c(xp, B, s);                                               // _resampler
        launch_scale_c(m1, m->n1m, c1m, 0.5f, B, s);                                // _amplify
        launch_scale_c(m1, m->n1m, c1m, m->bb_gain, B, s);                          // _bb_gain
        FirCccParams ff{}; ff.in = m1; ff.out = m2; ff.q0 = m->n1m; ff.count = c1m; ff.taps = m->t_chan.p; ff.nt = m->n_chan;
        ff.port = mod_out; ff.port_cap = mod_stride;
        launch_an_fir_ccc(ff, B, s);                                                // _filter: straight into the caller's buffer (or the back end's)
        back_end(c1m);
        HIPCHK(hipGetLastError());
        if (qrl::take_launch_error()) return QRL_ERR_HIP;
        m->n8 += c8; m->n1m += c1m; m->last = (size_t)c1m * m->be_interp;
        return QRL_OK;
    }
    if (m->ssb) {
        RingF a0{m->a0.p, m->m8}, a1{m->a1.p, m->m8};
        RingC c1{m->c1.p, m->m8}, c2{m->c2.p, m->m8}, c3{m->c3.p, m->m8};
        const uint32_t c8 = (uint32_t)n;
        const uint64_t n8_1 = m->n8 + n;
        const uint64_t ns_1 = n8_1 >= 2 ? 1024 * ((n8_1 - 2) / 1024) : 0;        // stretcher: whole chunks, two items of look-ahead
        const uint32_t cs = (uint32_t)(ns_1 - m->ns);
        if ((size_t)cs * m->sps * m->be_interp > out_stride) return qrl_set_error(QRL_ERR_ARG, "amod: out_stride smaller than this call's output (qrl_amod_out_cap)");
        if (m->cw) {   // the key's tone source instead of the caller's audio (which is ignored): amplitude 0.001 / 0.98 by qrl_amod_set_cw_k, offset 1
            AmToneParams tp{}; tp.out = a0; tp.n0 = m->n8; tp.count = c8; tp.tab = m->tone_tab.p; tp.inc = m->tone_inc; tp.k0 = m->cw_k; tp.ampl = m->cw_ampl; tp.offset = 1.0f;
            launch_am_tone(tp, B, s);
        } else {
            AmLoadParams lp{}; lp.in = audio; lp.in_stride = stride; lp.out = a0; lp.n0 = m->n8; lp.count = c8;
            launch_am_load(lp, B, s);
        }
        FirFffParams af{}; af.in = a0; af.out = a1; af.q0 = m->n8; af.count = c8; af.taps = m->t_audio.p; af.nt = m->n_audio;
        launch_fir_fff(af, B, s);                                                   // _audio_filter
        AmClipParams cp{}; cp.in = a1; cp.out = c1; cp.n0 = m->n8; cp.count = c8; cp.clip = 0.95f; cp.atan_tab = m->atan_tab.p;
        launch_am_clip(cp, B, s);                                                   // _float_to_complex, _clipper
        AmStretchParams sp{}; sp.in = c1; sp.out = c2; sp.q0 = m->ns; sp.count = cs;
        launch_am_stretch(sp, B, s);                                                // _stretcher
        FirCccParams ff{}; ff.in = c2; ff.out = c3; ff.q0 = m->ns; ff.count = cs; ff.taps = m->t_side.p; ff.nt = m->n_side;
        launch_an_fir_ccc(ff, B, s);                                                // _filter_usb / _filter_lsb
        launch_scale_c(c3, m->ns, cs, 0.9f, B, s);                                  // _amplify
        launch_scale_c(c3, m->ns, cs, m->bb_gain, B, s);                            // _bb_gain
        TxInterpCParams xp{}; xp.in = c3; xp.n0 = m->ns * (uint64_t)m->sps; xp.count = cs * (uint32_t)m->sps;
        xp.taps = m->t_interp.p; xp.nt = m->n_interp; xp.interp = m->sps; xp.out = mod_out; xp.out_stride = mod_stride;
        if (xp.count) launch_tx_interp_c(xp, B, s);                                 // _resampler
        back_end(xp.count);
        HIPCHK(hipGetLastError());
        if (qrl::take_launch_error()) return QRL_ERR_HIP;
        m->n8 = n8_1; m->ns = ns_1; m->last = (size_t)cs * m->sps * m->be_interp;
        if (m->cw) m->cw_k += c8;
        return QRL_OK;
    }
    RingF a0{m->a0.p, m->m8}, a1{m->a1.p, m->m8}, a2{m->a2.p, m->m8}, r50{m->r50.p, m->m50};
    RingC fmv{m->fmv.p, m->m50}, flt{m->flt.p, m->m50};
    const uint32_t c8 = (uint32_t)n, c50 = (uint32_t)(n * 25 / 4);
    if ((size_t)c50 * m->sps * m->be_interp > out_stride) return qrl_set_error(QRL_ERR_ARG, "amod: out_stride smaller than this call's output (qrl_amod_out_cap)");
    AmLoadParams lp{}; lp.in = audio; lp.in_stride = stride; lp.out = a0; lp.n0 = m->n8; lp.count = c8;
    launch_am_load(lp, B, s);
    const bool tone_on = m->tone_hz > 0.0f;
    FirFffParams af{}; af.in = a0; af.out = a1; af.q0 = m->n8; af.count = c8; af.taps = tone_on ? m->t_audio_bp.p : m->t_audio.p; af.nt = tone_on ? m->n_audio_bp : m->n_audio;
    launch_fir_fff(af, B, s);                                                       // _audio_filter (set_ctcss(tone): the band-pass)
    AmIirParams ip{}; ip.in = a1; ip.out = a2; ip.n0 = m->n8; ip.count = c8; ip.gain = m->k_audio;   // _audio_amplify, [_add tone], _pre_emph_filter
    if (tone_on) { ip.tone_tab = m->tone_tab.p; ip.tone_inc = m->tone_inc; ip.tone_k0 = m->tone_k; ip.tone_ampl = 0.15; m->tone_k += c8; }
    ip.ff0 = m->pb[0]; ip.ff1 = m->pb[1]; ip.fb1 = -m->pa[1]; ip.st = m->iir.p;
    launch_am_iir(ip, B, s);
    AnResampParams rp{}; rp.in = a2; rp.out = r50; rp.st = nullptr; rp.taps = m->t_if.p; rp.nt = m->n_if; rp.I = 25; rp.D = 4;
    rp.q0 = m->n50; rp.count = c50;
    launch_an_resamp(rp, c50, B, s);                                                // _if_resampler
    TxFmParams fp{}; fp.in = r50; fp.out = fmv; fp.n0 = m->n50; fp.count = c50; fp.k = m->fm_k; fp.amp = 1.0f; fp.phase = m->phase.p;
    launch_tx_fm(fp, B, s);                                                         // _fm_modulator
    FirCcfParams cf{}; cf.in = fmv; cf.out = flt; cf.q0 = m->n50; cf.count = c50; cf.taps = m->t_filt.p; cf.nt = m->n_filt;
    launch_fir_ccf(cf, B, s);                                                       // _filter
    launch_scale_c(flt, m->n50, c50, 0.8f, B, s);                                   // _amplify
    launch_scale_c(flt, m->n50, c50, m->bb_gain, B, s);                             // _bb_gain
    TxInterpCParams xp{}; xp.in = flt; xp.n0 = m->n50 * (uint64_t)m->sps; xp.count = c50 * (uint32_t)m->sps;
    xp.taps = m->t_interp.p; xp.nt = m->n_interp; xp.interp = m->sps; xp.out = mod_out; xp.out_stride = mod_stride;
    launch_tx_interp_c(xp, B, s);                                                   // _resampler
    back_end(xp.count);
    HIPCHK(hipGetLastError());
    if (qrl::take_launch_error()) return QRL_ERR_HIP;
    m->n8 += c8; m->n50 += c50; m->last = (size_t)c50 * m->sps * m->be_interp;
    return QRL_OK;
}

}
