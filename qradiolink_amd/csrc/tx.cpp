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
    int sps = 4;
    float bb_gain = 1.0f;
    float* taps = nullptr; int nt = 0;
    TxState* st = nullptr;
    uint8_t* sym = nullptr; uint32_t sym_mask = 0;
    uint64_t nsym = 0;   // symbols (= input bits) so far
    ~qrl_mod() {
        if (taps) (void)hipFree(taps);
        if (st) (void)hipFree(st);
        if (sym) (void)hipFree(sym);
        if (own_stream && stream) (void)hipStreamDestroy(stream);
    }
    int init_state() {
        std::vector<TxState> s(cfg.batch);
        for (auto& x : s) { x.sr = 0x7F; x.enc = 0; x.prev = 0; x.pad = 0; }   // scrambler seed 0x7F (gr_mod_qpsk.cpp:62)
        if (hipMemcpy(st, s.data(), s.size() * sizeof(TxState), hipMemcpyHostToDevice) != hipSuccess) return QRL_ERR_HIP;
        if (hipMemset(sym, 0, (size_t)cfg.batch * (sym_mask + 1)) != hipSuccess) return QRL_ERR_HIP;
        nsym = 0;
        return QRL_OK;
    }
};

// zero-input transition of scrambler_bb(0x8A, -, 7) applied L times, as 8 column masks
static void lfsr_power(uint32_t L, uint8_t cols[8])
{
    for (int k = 0; k < 8; ++k) {
        uint32_t sr = 1u << k;
        for (uint32_t i = 0; i < L; ++i) {
            const uint32_t nb = (uint32_t)__builtin_parity(sr & 0x8Au);
            sr = (sr >> 1) | (nb << 7);
        }
        cols[k] = (uint8_t)sr;
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
    if (c.use_mode_defaults) {
        if (c.modem_type != QRL_MODEM_QPSK250K) return qrl_set_error(QRL_ERR_ARG, "modulator: modem_type not supported by this build");
        c.sps = 4; c.samp_rate = 1000000; c.carrier_freq = 1700; c.filter_width = 160000;   // gr_mod_base.cpp:175
    }
    if (c.modem_type != QRL_MODEM_QPSK250K || c.sps < 2 || c.sps > 10)
        return qrl_set_error(QRL_ERR_ARG, "modulator: only the QPSK sps <= 10 geometry is built");
    m->sps = c.sps;
    m->bb_gain = c.bb_gain == 0.0f ? 1.0f : c.bb_gain;
    HIPCHK(hipSetDevice(ctx->device));
    if (c.hip_stream) m->stream = static_cast<hipStream_t>(c.hip_stream);
    else { HIPCHK(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking)); m->own_stream = true; }
    const std::vector<float> rrc = root_raised_cosine(m->sps, m->sps, 1, 0.35, 15 * m->sps);   // nfilts = 15 for sps <= 10
    m->nt = (int)rrc.size();
    if (m->nt > 256) return qrl_set_error(QRL_ERR_ARG, "modulator: pulse-shaping filter too long");
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&m->taps), rrc.size() * sizeof(float)));
    HIPCHK(hipMemcpy(m->taps, rrc.data(), rrc.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&m->st), (size_t)c.batch * sizeof(TxState)));
    uint32_t cap = 1024;
    while (cap < c.max_bytes * 8 + 256) cap <<= 1;
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
int qrl_mod_set_bb_gain(qrl_mod* m, float g) { if (!m) return QRL_ERR_ARG; m->bb_gain = g; return QRL_OK; }
size_t qrl_mod_samples_per_byte(const qrl_mod* m) { return m ? (size_t)8 * m->sps : 0; }

int qrl_mod_process(qrl_mod* m, const uint8_t* bytes, size_t stride, size_t nbytes, float* iq, size_t out_stride)
{
    if (!m || (!bytes && nbytes) || (!iq && nbytes)) return QRL_ERR_ARG;
    if (nbytes > m->cfg.max_bytes) return qrl_set_error(QRL_ERR_TOO_BIG, "nbytes exceeds max_bytes");
    if (nbytes == 0) return QRL_OK;
    HIPCHK(hipSetDevice(m->ctx->device));
    const int B = m->cfg.batch;
    const uint32_t nbits = (uint32_t)nbytes * 8;
    TxBitsParams p{};
    p.bytes = bytes; p.stride = stride; p.nbytes = (uint32_t)nbytes;
    p.L = ((nbits + 63) / 64 + 31) / 32 * 32;
    lfsr_power(p.L, p.tl_cols);
    p.st = m->st; p.sym = RingB{m->sym, m->sym_mask}; p.s0 = m->nsym;
    launch_tx_qpsk_bits(p, B, m->stream);
    TxInterpParams q{};
    q.sym = p.sym; q.n0 = m->nsym * (uint64_t)m->sps; q.count = nbits * (uint32_t)m->sps;
    q.taps = m->taps; q.nt = m->nt; q.interp = m->sps;
    // chunks_to_symbols_bc table of gr_mod_qpsk.cpp:44-54
    q.table[0] = make_float2(-0.707f, -0.707f); q.table[1] = make_float2(-0.707f, 0.707f);
    q.table[2] = make_float2(0.707f, 0.707f);   q.table[3] = make_float2(0.707f, -0.707f);
    q.amp = 0.6f; q.bb_gain = m->bb_gain;
    q.out = reinterpret_cast<float2*>(iq); q.out_stride = out_stride;
    launch_tx_interp(q, B, m->stream);
    HIPCHK(hipGetLastError());
    m->nsym += nbits;
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
