// tx.cpp — host side of the TX half of libqrl_hip.so: the "modulator" top_block of the reference
// (src/gr/gr_mod_base.cpp:25,175; src/gr/gr_mod_qpsk.cpp:56-89) as a two-kernel pipeline per call.
#include "../../include/qrl_hip.h"
#include "engine.hpp"
#include "firdes.hpp"
#include <hip/hip_runtime.h>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

using namespace qrl;

extern int qrl_set_error(int code, const std::string& msg);   // engine.cpp
struct qrl_ctx { int device; };

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return qrl_set_error(QRL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct qrl_mod {
    qrl_ctx* ctx = nullptr;
    qrl_mod_config cfg{};
    hipStream_t stream = nullptr;
    bool own_stream = false;
    enum { F_QPSK, F_FSK } fam = F_QPSK;   // F_QPSK: symbols -> RRC interpolator (QPSK, BPSK); F_FSK: shape -> FM -> interpolator
    bool bpsk = false, fsk4 = false; float shape_scale = 0.0f;
    // gr_mod_m17: raw dibits -> RRC x5 -> FM -> channel filter -> gains -> 125 / 3 (2500 samples per 3 bytes)
    bool m17 = false; float* m17_filt = nullptr; int m17_nf = 0; float2* m17_flt = nullptr;
    // gr_mod_dmr (src/gr/gr_mod_dmr.cpp:26-90): the m17 path with the DMR pulse and deviation; gr_zero_idle_bursts(62) in the place of the channel
    // filter = the stream 2 x 720 - 1 items late (the block's history, gr_zero_idle_bursts.cpp:34-37,76) + the tagged runs zeroed `delay` items early
    bool dmr = false; std::vector<ZeroRun> zero_runs; ZeroRun* zero_dev = nullptr; size_t zero_dev_cap = 0;
    static constexpr uint32_t kDmrHist = 2 * 720 - 1, kDmrTagDelay = 62;
    // gr_mod_dsss: coded bits -> Barker-13 chips -> RRC x25 (5200 sps) -> gains -> 50 / 13 (20 ksps) -> 1:50; 1 000 000 samples per byte
    bool dsss = false; uint8_t* ds_chips = nullptr; uint32_t ds_chip_mask = 0; float* ds_shaped = nullptr; float2 *ds_c52 = nullptr, *ds_c20 = nullptr;
    uint32_t ds_m52 = 0, ds_m20 = 0; float* ds_if_taps = nullptr; int ds_if_Jp = 0;
    int sps = 4;
    float bb_gain = 1.0f;
    float* taps = nullptr; int nt = 0;
    // FSK family (2FSK / GMSK): shaping taps (nt_shape = 0: repeat), FM constant, amplitude, second interpolator
    float* shape_taps = nullptr; int nt_shape = 0; float fm_k = 0, amplif = 0; int interp2 = 1;
    float* shaped = nullptr; float2* fmv = nullptr; uint32_t r1_mask = 0; float* phase = nullptr;
    TxState* st = nullptr;
    uint8_t* sym = nullptr; uint32_t sym_mask = 0;
    uint64_t nsym = 0;   // symbols (= input bits) so far
    // gr_mod_base back end (gr_mod_base.cpp:38,215-258): rotator at 1 Msps, then interpolation to the device rate
    bool backend = false; int be_interp = 1; float* be_taps = nullptr; int be_nt = 0;
    float2* bb = nullptr; size_t bb_stride = 0;            // modulator output, linear, one call's worth
    float2* be_ring = nullptr; uint32_t be_mask = 0;       // rotated 1 Msps signal (interpolator history)
    float2* rot_lo = nullptr; uint64_t rot_inc = 0, rot_acc = 0, rot_nbase = 0, n_bb = 0;
    int set_rot(double hz) {
        rot_inc = phase_inc_to_turn(2 * M_PI * hz / 1000000.0);
        std::vector<float2> lo(512);
        for (int r = 0; r < 512; ++r) { float s, c; sincos_turn_host((uint64_t)r * rot_inc, s, c); lo[r] = make_float2(c, s); }
        return hipMemcpy(rot_lo, lo.data(), 512 * sizeof(float2), hipMemcpyHostToDevice) == hipSuccess ? QRL_OK : QRL_ERR_HIP;
    }
    ~qrl_mod() {
        if (taps) (void)hipFree(taps);
        for (void* p : {(void*)zero_dev, (void*)shape_taps, (void*)shaped, (void*)fmv, (void*)phase, (void*)m17_filt, (void*)m17_flt, (void*)ds_chips, (void*)ds_shaped, (void*)ds_c52, (void*)ds_c20, (void*)ds_if_taps}) if (p) (void)hipFree(p);
        if (st) (void)hipFree(st);
        for (void* p : {(void*)be_taps, (void*)bb, (void*)be_ring, (void*)rot_lo}) if (p) (void)hipFree(p);
        if (sym) (void)hipFree(sym);
        if (own_stream && stream) (void)hipStreamDestroy(stream);
    }
    int init_state() {
        std::vector<TxState> s(cfg.batch);
        for (auto& x : s) { x.sr = 0x7F; x.enc = 0; x.prev = 0; x.pad = 0; }   // scrambler seed 0x7F (gr_mod_qpsk.cpp:62)
        if (hipMemcpy(st, s.data(), s.size() * sizeof(TxState), hipMemcpyHostToDevice) != hipSuccess) return QRL_ERR_HIP;
        if (hipMemset(sym, 0, (size_t)cfg.batch * (sym_mask + 1)) != hipSuccess) return QRL_ERR_HIP;
        if (fam == F_FSK) {
            if (hipMemset(shaped, 0, (size_t)cfg.batch * (r1_mask + 1) * sizeof(float)) != hipSuccess) return QRL_ERR_HIP;
            if (hipMemset(fmv, 0, (size_t)cfg.batch * (r1_mask + 1) * sizeof(float2)) != hipSuccess) return QRL_ERR_HIP;
            if (hipMemset(phase, 0, (size_t)cfg.batch * sizeof(float)) != hipSuccess) return QRL_ERR_HIP;
            if (m17_flt && hipMemset(m17_flt, 0, (size_t)cfg.batch * (r1_mask + 1) * sizeof(float2)) != hipSuccess) return QRL_ERR_HIP;
        }
        if (dsss) {
            if (hipMemset(ds_chips, 0, (size_t)cfg.batch * (ds_chip_mask + 1)) != hipSuccess) return QRL_ERR_HIP;
            if (hipMemset(ds_shaped, 0, (size_t)cfg.batch * (ds_m52 + 1) * sizeof(float)) != hipSuccess) return QRL_ERR_HIP;
            if (hipMemset(ds_c52, 0, (size_t)cfg.batch * (ds_m52 + 1) * sizeof(float2)) != hipSuccess) return QRL_ERR_HIP;
            if (hipMemset(ds_c20, 0, (size_t)cfg.batch * (ds_m20 + 1) * sizeof(float2)) != hipSuccess) return QRL_ERR_HIP;
        }
        if (be_ring && hipMemset(be_ring, 0, (size_t)cfg.batch * (be_mask + 1) * sizeof(float2)) != hipSuccess) return QRL_ERR_HIP;
        nsym = 0; n_bb = 0; rot_acc = 0; rot_nbase = 0;
        zero_runs.clear();
        return QRL_OK;
    }
};

// zero-input transition of scrambler_bb(0x8A, -, 7) applied L, 2 L, 4 L .. 32 L times, as 8 column masks each (the powers by squaring)
static void lfsr_powers(uint32_t L, uint8_t pow[6][8])
{
    for (int k = 0; k < 8; ++k) {
        uint32_t sr = 1u << k;
        for (uint32_t i = 0; i < L; ++i) {
            const uint32_t nb = (uint32_t)__builtin_parity(sr & 0x8Au);
            sr = (sr >> 1) | (nb << 7);
        }
        pow[0][k] = (uint8_t)sr;
    }
    for (int d = 1; d < 6; ++d)
        for (int k = 0; k < 8; ++k) {                  // column k of A^2 = A applied to column k of A
            uint32_t r = 0;
            for (int j = 0; j < 8; ++j) if ((pow[d - 1][k] >> j) & 1u) r ^= pow[d - 1][j];
            pow[d][k] = (uint8_t)r;
        }
}

extern "C" {

int qrl_mod_create(qrl_ctx* ctx, const qrl_mod_config* cfg, qrl_mod** outp)
{
    if (!ctx || !cfg || !outp) return QRL_ERR_ARG;
    if (cfg->batch < 1 || cfg->max_bytes < 1) return qrl_set_error(QRL_ERR_ARG, "batch and max_bytes must be >= 1");
    std::unique_ptr<qrl_mod> m(new (std::nothrow) qrl_mod);
    if (!m) return QRL_ERR_NOMEM;
    m->ctx = ctx; m->cfg = *cfg;
    qrl_mod_config& c = m->cfg;
    bool fsk = false, gmsk = false, fsk4 = false, bpsk = false;
    if (c.use_mode_defaults) {   // literals of gr_mod_base.cpp:154-175
        c.samp_rate = 1000000; c.carrier_freq = 1700; c.fm = 0;
        switch (c.modem_type) {
        case QRL_MODEM_QPSK250K:  c.sps = 4;   c.filter_width = 160000; break;
        case QRL_MODEM_QPSKVIDEO: c.sps = 4;   c.filter_width = 160000; break;            // gr_mod_base.cpp:176
        case QRL_MODEM_QPSK2K:    c.sps = 500; c.filter_width = 1300;  break;             // :173
        case QRL_MODEM_QPSK20K:   c.sps = 100; c.filter_width = 6500;  break;             // :174
        case QRL_MODEM_2FSK2KFM:  c.sps = 25;  c.filter_width = 4000;  c.fm = 1; break;
        case QRL_MODEM_2FSK1KFM:  c.sps = 50;  c.filter_width = 2500;  c.fm = 1; break;
        case QRL_MODEM_2FSK2K:    c.sps = 25;  c.filter_width = 4000;  break;
        case QRL_MODEM_2FSK1K:    c.sps = 50;  c.filter_width = 2000;  break;
        case QRL_MODEM_2FSK10KFM: c.sps = 5;   c.filter_width = 25000; c.fm = 1; break;
        case QRL_MODEM_GMSK2K:    c.sps = 50;  c.filter_width = 4000;  break;
        case QRL_MODEM_GMSK1K:    c.sps = 100; c.filter_width = 2000;  break;
        case QRL_MODEM_GMSK10K:   c.sps = 10;  c.filter_width = 20000; break;
        case QRL_MODEM_4FSK2K:    c.sps = 25;  c.filter_width = 4000;  break;             // gr_mod_base.cpp:163 (non-FM: repeat, spacing 2)
        case QRL_MODEM_4FSK2KFM:  c.sps = 25;  c.filter_width = 3500;  c.fm = 1; break;   // gr_mod_base.cpp:164
        case QRL_MODEM_4FSK1KFM:  c.sps = 50;  c.filter_width = 2000;  c.fm = 1; break;   // :165
        case QRL_MODEM_4FSK10KFM: c.sps = 5;   c.filter_width = 20000; c.fm = 1; break;   // :166
        case QRL_MODEM_4FSK100K:  c.sps = 2;   c.filter_width = 125000; c.fm = 1; break;  // :177
        case QRL_MODEM_BPSK1K:    c.sps = 500; c.filter_width = 1500;  break;             // :168
        case QRL_MODEM_BPSK2K:    c.sps = 250; c.filter_width = 2800;  break;             // :169
        case QRL_MODEM_M17:       c.sps = 125; c.filter_width = 9000;  break;             // make_gr_mod_m17() :206, defaults gr_mod_m17.h:43-44
        case QRL_MODEM_DMR:       c.sps = 125; c.filter_width = 5000;  break;             // make_gr_mod_dmr() :207, defaults of make_gr_mod_dmr gr_mod_dmr.h:38-39
        case QRL_MODEM_BPSK8:     c.sps = 25;  c.filter_width = 150;   break;             // make_gr_mod_dsss(25, 1000000, 1700, 150) :170
        default: return qrl_set_error(QRL_ERR_ARG, "modulator: modem_type not supported by this build");
        }
    }
    switch (c.modem_type) {
    case QRL_MODEM_QPSK250K: case QRL_MODEM_QPSKVIDEO: case QRL_MODEM_QPSK2K: case QRL_MODEM_QPSK20K: break;
    case QRL_MODEM_2FSK2KFM: case QRL_MODEM_2FSK1KFM: case QRL_MODEM_2FSK2K: case QRL_MODEM_2FSK1K: case QRL_MODEM_2FSK10KFM: fsk = true; break;
    case QRL_MODEM_GMSK2K: case QRL_MODEM_GMSK1K: case QRL_MODEM_GMSK10K: fsk = gmsk = true; break;
    case QRL_MODEM_4FSK2K: case QRL_MODEM_4FSK2KFM: case QRL_MODEM_4FSK1KFM: case QRL_MODEM_4FSK10KFM: case QRL_MODEM_4FSK100K: fsk = fsk4 = true; break;
    case QRL_MODEM_BPSK1K: case QRL_MODEM_BPSK2K: bpsk = true; break;
    case QRL_MODEM_M17: fsk = true; m->m17 = true; break;
    case QRL_MODEM_DMR: fsk = true; m->m17 = true; m->dmr = true; break;
    case QRL_MODEM_BPSK8: m->dsss = true; break;
    default: return qrl_set_error(QRL_ERR_ARG, "modulator: modem_type not supported by this build");
    }
    m->bpsk = bpsk; m->fsk4 = fsk4;
    if (m->m17 && c.sps != 125) return qrl_set_error(QRL_ERR_ARG, "modulator: m17 / dmr sps must be 125 (rational_resampler_ccf(sps, 3), gr_mod_m17.cpp:64-66)");
    if ((m->m17 || m->dsss) && ((c.device_samp_rate != 0 && c.device_samp_rate != 1000000) || c.carrier_offset_hz != 0.0))
        return qrl_set_error(QRL_ERR_ARG, "modulator: the gr_mod_base back end is not built for m17 / dsss");
    if (m->dsss && c.sps != 25) return qrl_set_error(QRL_ERR_ARG, "modulator: dsss sps must be 25");
    if (m->dsss && c.max_bytes > 64) return qrl_set_error(QRL_ERR_TOO_BIG, "modulator: dsss makes 1 000 000 samples per byte: max_bytes <= 64");
    if (bpsk && (c.sps < 2 || c.sps > 1000)) return qrl_set_error(QRL_ERR_ARG, "modulator: bpsk sps out of range");
    if (!fsk && !bpsk && (c.sps < 2 || c.sps > 1000)) return qrl_set_error(QRL_ERR_ARG, "modulator: qpsk sps out of range");
    m->sps = c.sps;
    m->bb_gain = c.bb_gain == 0.0f ? 1.0f : c.bb_gain;
    HIPCHK(hipSetDevice(ctx->device));
    if (c.hip_stream) m->stream = static_cast<hipStream_t>(c.hip_stream);
    else {
        int r0;
        if (std::getenv("QRL_CU_TX")) { if ((r0 = qrl::create_role_stream(&m->stream, 0, "TX"))) return r0; }
        else HIPCHK(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
        m->own_stream = true;
    }
    auto upload = [](const std::vector<float>& v, float** dst) -> int {   // followed by 64 zeros: k_tx_interp_sym reads its I x J = 64 taps unguarded
        const size_t bytes = (v.size() + 64) * sizeof(float);
        if (hipMalloc(reinterpret_cast<void**>(dst), bytes) != hipSuccess) return QRL_ERR_NOMEM;
        if (hipMemset(*dst, 0, bytes) != hipSuccess) return QRL_ERR_HIP;
        if (!v.empty() && hipMemcpy(*dst, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return QRL_ERR_HIP;
        return QRL_OK;
    };
    size_t ring_items = c.max_bytes * 8 + 256;   // symbol ring: one item per input bit (QPSK) or per coded bit (FSK: x2)
    if (m->dsss) {   // gr_mod_dsss.cpp:60-76
        const int fw = c.filter_width;
        const std::vector<float> rrc = root_raised_cosine(25, 25, 1, 0.35, 11 * 25);                        // _resampler (25, 1)
        const std::vector<float> ti = low_pass(50.0, 5200.0 * 50, fw, fw * 5);                              // _resampler_if (50, 13)
        const std::vector<float> tr = low_pass(50, c.samp_rate, fw, fw * 5);                                // _resampler_rf (50, 1)
        m->nt_shape = (int)rrc.size(); m->nt = (int)tr.size();
        if (m->nt_shape > 1536) return qrl_set_error(QRL_ERR_ARG, "modulator: shaping filter too long");
        m->ds_if_Jp = ((int)ti.size() + 49) / 50;
        std::vector<float> lay((size_t)50 * m->ds_if_Jp, 0.0f);
        for (size_t k = 0; k < ti.size(); ++k) lay[(k % 50) * m->ds_if_Jp + k / 50] = ti[k];
        int r0;
        if ((r0 = upload(rrc, &m->shape_taps)) || (r0 = upload(tr, &m->taps)) || (r0 = upload(lay, &m->ds_if_taps))) return r0;
        m->shape_scale = 0.65f;                                                                             // _amplify
        ring_items = c.max_bytes * 16 + 256;                                                                // coded bits
        uint32_t cc = 1024, c52 = 1024, c20 = 1024;
        while (cc < c.max_bytes * 208 + 256) cc <<= 1;
        while (c52 < c.max_bytes * 5200 + 1024) c52 <<= 1;
        while (c20 < c.max_bytes * 20000 + 1024) c20 <<= 1;
        m->ds_chip_mask = cc - 1; m->ds_m52 = c52 - 1; m->ds_m20 = c20 - 1;
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&m->ds_chips), (size_t)c.batch * cc));
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&m->ds_shaped), (size_t)c.batch * c52 * sizeof(float)));
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&m->ds_c52), (size_t)c.batch * c52 * sizeof(float2)));
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&m->ds_c20), (size_t)c.batch * c20 * sizeof(float2)));
    } else if (!fsk) {
        const std::vector<float> rrc = bpsk ? root_raised_cosine(m->sps, m->sps, 1, 0.35, 11 * m->sps)   // gr_mod_bpsk.cpp:52-54
                                            : root_raised_cosine(m->sps, m->sps, 1, 0.35,             // gr_mod_qpsk.cpp:46-51
                                                                 (m->sps > 120 ? 11 : m->sps > 10 ? 13 : 15) * m->sps);
        m->nt = (int)rrc.size();
        if (m->nt > 16384) return qrl_set_error(QRL_ERR_ARG, "modulator: pulse-shaping filter too long");
        if (bpsk) ring_items = c.max_bytes * 16 + 256;
        int r0 = upload(rrc, &m->taps);
        if (r0) return r0;
    } else {
        m->fam = qrl_mod::F_FSK;
        int sps = c.sps, nfilts;
        std::vector<float> shape;
        if (m->m17) {   // gr_mod_m17.cpp:39-70 / gr_mod_dmr.cpp:36-62: five samples per symbol at 24 ksps
            sps = 5; m->interp2 = 1; m->amplif = 1.0f;
            shape = m->dmr ? root_raised_cosine(5, 24000, 4800, 0.2, 125) : root_raised_cosine(5, 5, 1, 0.5, 250); m->shape_scale = (float)0.66666666;
            m->fm_k = m->dmr ? (float)((M_PI * 4800.0 * 0.85) / 24000.0) : (float)(M_PI / 5);   // frequency_modulator_fc takes a float
            m->fsk4 = true;      // four levels per ring item
        } else if (gmsk) {   // gr_mod_gmsk.cpp:40-70
            nfilts = 35; m->interp2 = 5; m->amplif = 0.9f;
            if (sps == 10) { sps = 50; m->interp2 = 1; nfilts = 55; }
            if (sps == 50) nfilts = 55;
            if (sps == 100) nfilts = 35;
            if ((nfilts % 2) == 0) nfilts += 1;
            shape = gaussian(sps, sps, 0.3, nfilts);
            m->fm_k = (float)((M_PI / 2) / sps);
        } else if (fsk4) {   // gr_mod_4fsk.cpp:52-92
            nfilts = sps * 10; m->interp2 = 20;
            if (sps == 2) { sps = 5; m->interp2 = 2; nfilts = 256; }
            m->amplif = c.fm ? 0.9f : 0.8f;
            if (c.fm) { shape = root_raised_cosine(sps, sps, 1, 0.2, nfilts); m->shape_scale = (float)0.66666666; }
            m->fm_k = (float)(((c.fm ? 1 : 2) * M_PI) / sps);
        } else {      // gr_mod_2fsk.cpp:38-62
            nfilts = 25 * sps; m->interp2 = 10; m->amplif = c.fm ? 0.9f : 0.8f;
            if (sps == 5) nfilts *= 5;
            if ((nfilts % 2) == 0) nfilts += 1;
            if (c.fm) shape = root_raised_cosine(sps, sps, 1, 0.2, nfilts);
            m->fm_k = (float)(((c.fm ? 1 : 2) * M_PI / 2) / sps);
        }
        m->sps = sps;
        m->nt_shape = (int)shape.size();
        if (m->nt_shape > 1536) return qrl_set_error(QRL_ERR_ARG, "modulator: shaping filter too long");
        int r0 = upload(shape, &m->shape_taps);
        if (r0) return r0;
        const std::vector<float> lp = m->dmr ? low_pass_2(125, (double)c.samp_rate * 3, c.filter_width, 2000, 60, WIN_BLACKMAN_HARRIS)   // gr_mod_dmr.cpp:65-68
                                    : m->m17 ? low_pass(125, (double)c.samp_rate * 3, 12000, 12000, WIN_BLACKMAN_HARRIS)   // _resampler (125, 3), gr_mod_m17.cpp:64-66
                                             : low_pass(m->interp2, c.samp_rate, c.filter_width, c.filter_width, WIN_HAMMING);
        m->nt = (int)lp.size();
        if (m->nt > (m->dmr ? 8192 : 2048)) return qrl_set_error(QRL_ERR_ARG, "modulator: interpolator filter too long");   // (gr_mod_dmr: 4091 taps; k_tx_interp_c reads taps beyond 2048 through the caches)
        if ((r0 = upload(lp, &m->taps))) return r0;
        ring_items = c.max_bytes * 16 + 256;
        uint32_t cap1 = 1024;
        while (cap1 < c.max_bytes * 16 * (size_t)sps + (size_t)m->nt + 256 + (m->dmr ? qrl_mod::kDmrHist + 64 : 0)) cap1 <<= 1;
        m->r1_mask = cap1 - 1;
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&m->shaped), (size_t)c.batch * cap1 * sizeof(float)));
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&m->fmv), (size_t)c.batch * cap1 * sizeof(float2)));
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&m->phase), (size_t)c.batch * sizeof(float)));
        if (m->m17) {
            // _filter, gr_mod_m17.cpp:69-70.  gr_mod_dmr: one tap 1.0 at lag 2 x 720 - 1 -- the delay of gr_zero_idle_bursts' history read
            // through the same FIR kernel (fmaf(1, x, +0) = x; the zero taps leave the chain untouched)
            std::vector<float> ft = low_pass(1, 24000, c.filter_width, c.filter_width, WIN_BLACKMAN_HARRIS);
            if (m->dmr) { ft.assign(qrl_mod::kDmrHist + 1, 0.0f); ft.back() = 1.0f; }
            m->m17_nf = (int)ft.size();
            if ((r0 = upload(ft, &m->m17_filt))) return r0;
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&m->m17_flt), (size_t)c.batch * cap1 * sizeof(float2)));
        }
    }
    if (c.device_samp_rate != 0 && c.device_samp_rate != 1000000 &&
        (c.device_samp_rate < 2000000 || c.device_samp_rate % 1000000 != 0 || c.device_samp_rate > 64000000))
        return qrl_set_error(QRL_ERR_ARG, "modulator: device_samp_rate must be 1e6 or a multiple of 1e6 in [2e6, 64e6]");
    m->be_interp = c.device_samp_rate >= 2000000 ? c.device_samp_rate / 1000000 : 1;
    m->backend = m->be_interp > 1 || c.carrier_offset_hz != 0.0;
    if (m->backend) {
        const size_t spb1 = fsk4 ? (size_t)8 * m->sps * m->interp2 : fsk ? (size_t)16 * m->sps * m->interp2 : (size_t)(bpsk ? 16 : 8) * m->sps;
        m->bb_stride = c.max_bytes * spb1;
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&m->bb), (size_t)c.batch * m->bb_stride * sizeof(float2)));
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&m->rot_lo), 512 * sizeof(float2)));
        int r0 = m->set_rot(c.carrier_offset_hz);
        if (r0) return r0;
        if (m->be_interp > 1) {
            const std::vector<float> lp = low_pass(m->be_interp, c.device_samp_rate, 480000, 20000, WIN_BLACKMAN_HARRIS);
            m->be_nt = (int)lp.size();
            if ((r0 = upload(lp, &m->be_taps))) return r0;
            uint32_t capb = 1024;
            while (capb < m->bb_stride + (size_t)m->be_nt / m->be_interp + 64) capb <<= 1;
            m->be_mask = capb - 1;
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&m->be_ring), (size_t)c.batch * capb * sizeof(float2)));
        }
    }
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&m->st), (size_t)c.batch * sizeof(TxState)));
    uint32_t cap = 1024;
    while (cap < ring_items) cap <<= 1;
    m->sym_mask = cap - 1;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&m->sym), (size_t)c.batch * cap));
    int r = m->init_state();
    if (r) return r;
    *outp = m.release();
    return QRL_OK;
}
void qrl_mod_destroy(qrl_mod* m) { if (m) { (void)hipStreamSynchronize(m->stream); delete m; } }
int qrl_mod_reset(qrl_mod* m)
{
    if (!m) return QRL_ERR_ARG;
    HIPCHK(hipStreamSynchronize(m->stream));
    return m->init_state();
}
int qrl_mod_set_carrier_offset(qrl_mod* m, double hz)
{
    if (!m) return QRL_ERR_ARG;
    if (!m->backend) return qrl_set_error(QRL_ERR_ARG, "modulator was created without the gr_mod_base back end");
    HIPCHK(hipStreamSynchronize(m->stream));   // rot_lo is rewritten below
    m->rot_acc += (m->n_bb - m->rot_nbase) * m->rot_inc;   // phase-continuous, like rotator_cc::set_phase_inc
    m->rot_nbase = m->n_bb;
    return m->set_rot(hz);
}
int qrl_mod_set_bb_gain(qrl_mod* m, float g) { if (!m) return QRL_ERR_ARG; m->bb_gain = g; return QRL_OK; }
int qrl_mod_add_zero_runs(qrl_mod* m, const qrl_zero_run* runs, size_t n)
{
    if (!m || (!runs && n)) return QRL_ERR_ARG;
    if (!m->dmr) return qrl_set_error(QRL_ERR_ARG, "qrl_mod_add_zero_runs: QRL_MODEM_DMR handles only (gr_mod_dmr is the one modulator with a gr_zero_idle_bursts)");
    for (size_t i = 0; i < n; ++i) {
        if (runs[i].stream < 0 || runs[i].stream >= m->cfg.batch) return qrl_set_error(QRL_ERR_ARG, "zero run: stream out of range");
        if (runs[i].start < qrl_mod::kDmrTagDelay) continue;                            // `tag.offset == nitems + i + _delay` has no item to match (gr_zero_idle_bursts.cpp:64)
        // the counter is loaded at OUTPUT item T - delay; one counter per stream, a later tag overwrites it (as qrl_synth_add_zero_runs)
        ZeroRun z{(uint32_t)runs[i].stream, 0u, runs[i].start - qrl_mod::kDmrTagDelay, runs[i].count};
        for (ZeroRun& o : m->zero_runs) {
            if (o.row != z.row) continue;
            if (o.start < z.start && o.start + o.count > z.start) o.count = z.start - o.start;
            else if (z.start < o.start && z.start + z.count > o.start) z.count = o.start - z.start;
        }
        m->zero_runs.push_back(z);
    }
    return QRL_OK;
}
size_t qrl_mod_samples_per_block(const qrl_mod* m, size_t* bytes_per_block)
{
    if (!m) return 0;
    if (m->m17) { if (bytes_per_block) *bytes_per_block = 3; return 2500; }   // 4 symbols x 5 x 125 / 3 per byte
    if (m->dsss) { if (bytes_per_block) *bytes_per_block = 1; return 1000000; }
    if (bytes_per_block) *bytes_per_block = 1;
    return qrl_mod_samples_per_byte(m);
}
size_t qrl_mod_samples_per_byte(const qrl_mod* m)
{
    if (!m) return 0;
    if (m->m17) return 0;   // 833 1/3: see qrl_mod_samples_per_block
    if (m->dsss) return 1000000;   // 16 coded bits x 13 chips x 25 x 50 / 13 x 50
    const size_t spb1 = m->fsk4 ? (size_t)8 * m->sps * m->interp2 : m->fam == qrl_mod::F_FSK ? (size_t)16 * m->sps * m->interp2
                                : (size_t)(m->bpsk ? 16 : 8) * m->sps;
    return spb1 * (size_t)m->be_interp;
}

int qrl_mod_process(qrl_mod* m, const uint8_t* bytes, size_t stride, size_t nbytes, float* iq, size_t out_stride)
{
    if (!m || (!bytes && nbytes) || (!iq && nbytes)) return QRL_ERR_ARG;
    if (nbytes > m->cfg.max_bytes) return qrl_set_error(QRL_ERR_TOO_BIG, "nbytes exceeds max_bytes");
    if (nbytes == 0) return QRL_OK;
    HIPCHK(hipSetDevice(m->ctx->device));
    const int B = m->cfg.batch;
    const uint32_t nbits = (uint32_t)nbytes * 8;
    float2* mod_out = m->backend ? m->bb : reinterpret_cast<float2*>(iq);
    const size_t mod_stride = m->backend ? m->bb_stride : out_stride;
    auto back_end = [&](uint32_t n1) {   // n1 samples per stream at 1 Msps are in bb
        TxRotParams rp{}; rp.in = m->bb; rp.in_stride = m->bb_stride; rp.n0 = m->n_bb; rp.count = n1;
        rp.rot_acc = m->rot_acc; rp.rot_inc = m->rot_inc; rp.rot_nbase = m->rot_nbase; rp.rot_lo = m->rot_lo;
        if (m->be_interp > 1) rp.out_ring = RingC{m->be_ring, m->be_mask};
        else { rp.out = reinterpret_cast<float2*>(iq); rp.out_stride = out_stride; }
        launch_tx_rot(rp, B, m->stream);
        if (m->be_interp > 1) {
            TxInterpCParams bp{}; bp.in = rp.out_ring; bp.n0 = m->n_bb * (uint64_t)m->be_interp; bp.count = n1 * (uint32_t)m->be_interp;
            bp.taps = m->be_taps; bp.nt = m->be_nt; bp.interp = m->be_interp;
            bp.out = reinterpret_cast<float2*>(iq); bp.out_stride = out_stride;
            launch_tx_interp_c(bp, B, m->stream);
        }
        m->n_bb += n1;
    };
    if (m->m17) {
        if (nbytes % 3) return qrl_set_error(QRL_ERR_ARG, "modulator: m17 takes multiples of 3 bytes per call (2500 samples per 3 bytes)");
        const uint32_t nsy = (uint32_t)nbytes * 4, c24 = nsy * 5, cout = c24 / 3 * 125;
        RingB sym{m->sym, m->sym_mask};
        launch_tx_raw_dibits(bytes, stride, (uint32_t)nbytes, sym, m->nsym, B, m->stream);
        const uint64_t n24 = m->nsym * 5;
        TxShapeParams sp{}; sp.sym = sym; sp.out = RingF{m->shaped, m->r1_mask}; sp.n0 = n24; sp.count = c24; sp.sps = 5;
        sp.taps = m->shape_taps; sp.nt = m->nt_shape; sp.levels = 4; sp.scale = m->shape_scale;
        launch_tx_shape(sp, B, m->stream);                                              // _chunks_to_symbols, _first_resampler, _scale_pulses
        TxFmParams fp{}; fp.in = sp.out; fp.out = RingC{m->fmv, m->r1_mask}; fp.n0 = n24; fp.count = c24; fp.k = m->fm_k; fp.amp = 1.0f;
        fp.phase = m->phase;
        launch_tx_fm(fp, B, m->stream);                                                 // _fm_modulator
        RingC flt{m->m17_flt, m->r1_mask};
        FirCcfParams cf{}; cf.in = fp.out; cf.out = flt; cf.q0 = n24; cf.count = c24; cf.taps = m->m17_filt; cf.nt = m->m17_nf;
        launch_fir_ccf(cf, B, m->stream);                                               // _filter (gr_mod_dmr: the history delay of _zero_idle)
        if (m->dmr && !m->zero_runs.empty()) {                                          // _zero_idle: the tagged runs of this call's items
            const uint64_t lo = n24, hi = n24 + c24;
            std::vector<ZeroRun> live, keep;
            for (const ZeroRun& z : m->zero_runs) {
                if (z.start < hi && z.start + z.count > lo) live.push_back(z);
                if (z.start + z.count > hi) keep.push_back(z);
            }
            if (!live.empty()) {
                if (live.size() > m->zero_dev_cap) {
                    HIPCHK(hipStreamSynchronize(m->stream));
                    if (m->zero_dev) (void)hipFree(m->zero_dev);
                    m->zero_dev = nullptr; m->zero_dev_cap = 0;
                    HIPCHK(hipMalloc(reinterpret_cast<void**>(&m->zero_dev), live.size() * 2 * sizeof(ZeroRun)));
                    m->zero_dev_cap = live.size() * 2;
                }
                HIPCHK(hipMemcpyAsync(m->zero_dev, live.data(), live.size() * sizeof(ZeroRun), hipMemcpyHostToDevice, m->stream));
                HIPCHK(hipStreamSynchronize(m->stream));                                // (`live` is pageable host memory: the copy has left it)
                launch_zero_runs(flt, m->zero_dev, (uint32_t)live.size(), lo, hi, m->stream);
            }
            m->zero_runs.swap(keep);
        }
        launch_scale_c(flt, n24, c24, 0.9f, B, m->stream);                              // _amplify
        launch_scale_c(flt, n24, c24, m->bb_gain, B, m->stream);                        // _bb_gain
        TxInterpCParams ip{}; ip.in = flt; ip.n0 = n24 / 3 * 125; ip.count = cout; ip.taps = m->taps; ip.nt = m->nt; ip.interp = 125; ip.decim = 3;
        ip.out = reinterpret_cast<float2*>(iq); ip.out_stride = out_stride;
        launch_tx_interp_c(ip, B, m->stream);                                           // _resampler (125, 3)
        HIPCHK(hipGetLastError());
        if (qrl::take_launch_error()) return QRL_ERR_HIP;
        m->nsym += nsy;
        return QRL_OK;
    }
    TxBitsParams p{};
    p.bytes = bytes; p.stride = stride; p.nbytes = (uint32_t)nbytes;
    p.L = ((nbits + 63) / 64 + 31) / 32 * 32;
    lfsr_powers(p.L, p.tl_pow);
    p.st = m->st; p.sym = RingB{m->sym, m->sym_mask}; p.s0 = m->nsym;
    p.mode = m->fsk4 ? 2 : (m->fam == qrl_mod::F_FSK || m->bpsk || m->dsss) ? 1 : 0;
    launch_tx_qpsk_bits(p, B, m->stream);
    if (m->dsss) {
        const uint32_t ncoded = 2 * nbits, nchips = ncoded * 13u, c52 = nchips * 25u, c20 = c52 / 13u * 50u;
        const uint64_t chip0 = m->nsym * 13ull, n52 = chip0 * 25ull, n20 = n52 / 13ull * 50ull;
        RingB chips{m->ds_chips, m->ds_chip_mask};
        launch_tx_spread(p.sym, chips, m->nsym, ncoded, B, m->stream);                   // _dsss_encoder
        TxShapeParams sp{}; sp.sym = chips; sp.out = RingF{m->ds_shaped, m->ds_m52}; sp.n0 = n52; sp.count = c52; sp.sps = 25;
        sp.taps = m->shape_taps; sp.nt = m->nt_shape; sp.levels = 2; sp.scale = m->shape_scale;
        launch_tx_shape(sp, B, m->stream);                                               // _chunks_to_symbols, _resampler, _amplify
        RingC r52{m->ds_c52, m->ds_m52}, r20{m->ds_c20, m->ds_m20};
        launch_tx_f2c(sp.out, r52, n52, c52, m->bb_gain, B, m->stream);                  // _bb_gain
        ResampParams rp{}; rp.in = nullptr; rp.in_ring = r52; rp.n0 = n52; rp.n = c52; rp.out = r20; rp.q0 = n20; rp.q_count = c20;
        rp.taps = m->ds_if_taps; rp.I = 50; rp.D = 13; rp.Jp = m->ds_if_Jp;
        launch_resamp(rp, B, m->stream);                                                 // _resampler_if (50, 13)
        TxInterpCParams ip{}; ip.in = r20; ip.n0 = n20 * 50ull; ip.count = c20 * 50u; ip.taps = m->taps; ip.nt = m->nt; ip.interp = 50;
        ip.out = reinterpret_cast<float2*>(iq); ip.out_stride = out_stride;
        launch_tx_interp_c(ip, B, m->stream);                                            // _resampler_rf (50, 1)
        HIPCHK(hipGetLastError());
        if (qrl::take_launch_error()) return QRL_ERR_HIP;
        m->nsym += ncoded;
        return QRL_OK;
    }
    if (m->fam == qrl_mod::F_FSK) {
        // nsym counts CODED bits here (2 per input bit); rate-1 samples = coded bits * sps
        const uint32_t ncoded = m->fsk4 ? nbits : 2 * nbits;   // ring items per call: 4-level symbols, or coded bits
        const uint64_t n1_0 = m->nsym * (uint64_t)m->sps;
        const uint32_t c1 = ncoded * (uint32_t)m->sps;
        TxShapeParams sp{}; sp.sym = p.sym; sp.out = RingF{m->shaped, m->r1_mask}; sp.n0 = n1_0; sp.count = c1; sp.sps = m->sps;
        sp.taps = m->shape_taps; sp.nt = m->nt_shape; sp.levels = m->fsk4 ? 4 : 2; sp.scale = m->shape_scale;
        launch_tx_shape(sp, B, m->stream);
        TxFmParams fp{}; fp.in = sp.out; fp.out = RingC{m->fmv, m->r1_mask}; fp.n0 = n1_0; fp.count = c1; fp.k = m->fm_k; fp.amp = m->amplif;
        fp.phase = m->phase;
        launch_tx_fm(fp, B, m->stream);
        TxInterpCParams ip{}; ip.in = fp.out; ip.n0 = n1_0 * (uint64_t)m->interp2; ip.count = c1 * (uint32_t)m->interp2;
        ip.taps = m->taps; ip.nt = m->nt; ip.interp = m->interp2; ip.out = mod_out; ip.out_stride = mod_stride;
        launch_tx_interp_c(ip, B, m->stream);
        if (m->backend) back_end(ip.count);
        HIPCHK(hipGetLastError());
    if (qrl::take_launch_error()) return QRL_ERR_HIP;
        m->nsym += ncoded;
        return QRL_OK;
    }
    TxInterpParams q{};
    const uint32_t nitems = m->bpsk ? 2 * nbits : nbits;   // BPSK: one symbol per coded bit (gr_mod_bpsk.cpp:60-61)
    q.sym = p.sym; q.n0 = m->nsym * (uint64_t)m->sps; q.count = nitems * (uint32_t)m->sps;
    q.taps = m->taps; q.nt = m->nt; q.interp = m->sps;
    // chunks_to_symbols_bc table of gr_mod_qpsk.cpp:44-54
    q.table[0] = make_float2(-0.707f, -0.707f); q.table[1] = make_float2(-0.707f, 0.707f);
    q.table[2] = make_float2(0.707f, 0.707f);   q.table[3] = make_float2(0.707f, -0.707f);
    if (m->bpsk) { q.table[0] = make_float2(-1.0f, 0.0f); q.table[1] = make_float2(1.0f, 0.0f); }   // gr_mod_bpsk.cpp:33-35
    q.amp = 0.6f; q.bb_gain = m->bb_gain;
    q.out = mod_out; q.out_stride = mod_stride;
    launch_tx_interp(q, B, m->stream);
    if (m->backend) back_end(q.count);
    HIPCHK(hipGetLastError());
    if (qrl::take_launch_error()) return QRL_ERR_HIP;
    m->nsym += nitems;
    return QRL_OK;
}
int qrl_mod_sync(qrl_mod* m)
{
    if (!m) return QRL_ERR_ARG;
    HIPCHK(hipStreamSynchronize(m->stream));
    return QRL_OK;
}
void* qrl_mod_stream(qrl_mod* m) { return m ? m->stream : nullptr; }

}  // extern "C"
