/* orc_pipeline.c -- TEST INFRASTRUCTURE (bench.py's cpu_baseline leg only; never part of the product path).
 *
 * SURVEY.md 8(d), CPU baseline variant (ii): GNU Radio's thread-per-block scheduler.  The reference's flowgraph runs every block on a
 * thread of its own and the streams flow from block to block through buffers (gnuradio-runtime tpb_thread_body; the 2FSK receiver of
 * gr_demod_2fsk.cpp:82-164 behind gr_demod_base's rotator is eleven such blocks).  Rounds 3-5 MODELLED that figure as samples / run time
 * of the slowest block; this file EMULATES it: the same restated blocks (oracle/orc_blocks.c, called exactly as orc_demod_2fsk calls them),
 * one POSIX thread per block, bounded single-producer single-consumer queues in between.  The unit that flows is one stream's buffer of a
 * call (GNU Radio moves smaller chunks; the steady-state rate of a pipeline does not depend on the chunk size once the queues are full).
 * Result = the descrambled bits of every stream, summed into the same checksum orc_batch_rx computes: tests/test_oracle_pipeline.py holds
 * the two equal.  Filters are designed once, as the reference designs them in the constructor. */
#include "orc.h"
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define NEW(T, n) ((T*)calloc((size_t)(n) + 16, sizeof(T)))
#define NSTAGE 11
#define QCAP 4

typedef struct {
    const cf32* in; size_t n;
    cf32 *rot, *s1, *s2, *filt, *fu, *fl;
    float *s4, *s5, *sym;
    uint8_t *soft, *bits_a, *bits_b;
    size_t n1, nsym, nba, nbb;
} job_t;

typedef struct {
    job_t* slot[QCAP];
    int head, tail, count;
    pthread_mutex_t mu; pthread_cond_t cv;
} queue_t;

static void q_init(queue_t* q) { memset(q, 0, sizeof *q); pthread_mutex_init(&q->mu, NULL); pthread_cond_init(&q->cv, NULL); }
static void q_put(queue_t* q, job_t* j)
{
    pthread_mutex_lock(&q->mu);
    while (q->count == QCAP) pthread_cond_wait(&q->cv, &q->mu);
    q->slot[q->tail] = j; q->tail = (q->tail + 1) % QCAP; q->count++;
    pthread_cond_broadcast(&q->cv);
    pthread_mutex_unlock(&q->mu);
}
static job_t* q_get(queue_t* q)
{
    pthread_mutex_lock(&q->mu);
    while (q->count == 0) pthread_cond_wait(&q->cv, &q->mu);
    job_t* j = q->slot[q->head]; q->head = (q->head + 1) % QCAP; q->count--;
    pthread_cond_broadcast(&q->cv);
    pthread_mutex_unlock(&q->mu);
    return j;
}

typedef struct {
    /* the constructor's designs (gr_demod_2fsk.cpp:55-80 for sps = 10, samp_rate 1e6) */
    uint64_t rot_inc;
    float *rs; int nrs;           /* _resampler 1:50 */
    float *ft; int nft;           /* _filter */
    cf32 *up, *lo; int nb;        /* _upper_filter / _lower_filter */
    float *st; int nst;           /* _symbol_filter */
    queue_t q[NSTAGE + 1];
    double busy[NSTAGE];
} pipe_t;

typedef struct { pipe_t* p; int stage; } targ_t;

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

static void run_stage(pipe_t* P, int k, job_t* j)
{
    const int target = 20000, sps_eff = 10;
    switch (k) {
    case 0:   /* rotator_cc (gr_demod_base.cpp:84-86) */
        j->rot = NEW(cf32, j->n);
        orc_rotator(j->in, j->n, P->rot_inc, 0, j->rot);
        break;
    case 1:   /* rational_resampler_ccf(1, 50) */
        j->n1 = orc_decim_count(j->n, 1, 50);
        j->s1 = NEW(cf32, j->n1);
        orc_decim_auto(j->rot, j->n, P->rs, P->nrs, 50, j->s1);
        free(j->rot); j->rot = NULL;
        break;
    case 2:   /* fll_band_edge_cc */
        j->s2 = NEW(cf32, j->n1);
        orc_fll_band_edge(j->s1, j->n1, (float)sps_eff, 0.1f, 16, (float)(24 * M_PI / 100), j->s2);
        free(j->s1); j->s1 = NULL;
        break;
    case 3:   /* fft_filter_ccf (_filter) */
        j->filt = NEW(cf32, j->n1);
        orc_fir_ccf(j->s2, j->n1, P->ft, P->nft, j->filt);
        free(j->s2); j->s2 = NULL;
        break;
    case 4:   /* _upper_filter + _lower_filter (two fft_filter_ccc blocks; one stage here, as the oracle computes them together) */
        j->fu = NEW(cf32, j->n1); j->fl = NEW(cf32, j->n1);
        orc_fir_ccc_conj_pair(j->filt, j->n1, P->up, P->lo, P->nb, j->fu, j->fl);
        free(j->filt); j->filt = NULL;
        break;
    case 5:   /* complex_to_mag x 2, divide_ff, rail_ff, add_const_ff */
        j->s4 = NEW(float, j->n1);
        for (size_t i = 0; i < j->n1; i++) {
            float mu = sqrtf(j->fu[i].re * j->fu[i].re + j->fu[i].im * j->fu[i].im);
            float ml = sqrtf(j->fl[i].re * j->fl[i].re + j->fl[i].im * j->fl[i].im);
            float r = mu / ml;
            if (!(r >= 0.0f)) r = 0.0f;
            if (r > 2.0f) r = 2.0f;
            j->s4[i] = r + (-1.0f);
        }
        free(j->fu); free(j->fl); j->fu = j->fl = NULL;
        break;
    case 6:   /* _symbol_filter */
        j->s5 = NEW(float, j->n1);
        orc_fir_fff(j->s4, j->n1, P->st, P->nst, j->s5);
        free(j->s4); j->s4 = NULL;
        break;
    case 7: { /* symbol_sync_ff */
        const float symbol_rate = (float)target / (float)sps_eff;
        j->sym = NEW(float, j->n1 / (size_t)(sps_eff - 1) + 16);
        j->nsym = orc_symbol_sync_ff(j->s5, j->n1, ORC_TED_MOD_MM, (float)sps_eff, (float)(2 * M_PI / (symbol_rate / 10)), 1.0f, 0.2869f,
                                     200.0f / symbol_rate, ORC_CONST_BPSK, j->sym);
        free(j->s5); j->s5 = NULL;
        break;
    }
    case 8:   /* multiply_const, add_const, float_to_uchar; delay(1) for the second branch */
        j->soft = NEW(uint8_t, j->nsym + 1);
        orc_soft_quant(j->sym, j->nsym, 128.0f, 128.0f, j->soft + 1);
        j->soft[0] = 0;
        free(j->sym); j->sym = NULL;
        break;
    case 9: { /* fec::decoder + descrambler, branch A */
        uint8_t* dec = NEW(uint8_t, j->nsym / 2 + 80);
        j->nba = orc_cc_decode_k7(j->soft + 1, j->nsym, dec);
        j->bits_a = NEW(uint8_t, j->nba);
        orc_descramble(dec, j->nba, 0x8A, 0x7F, 7, j->bits_a);
        free(dec);
        break;
    }
    case 10: { /* branch B (the delayed stream) */
        uint8_t* dec = NEW(uint8_t, j->nsym / 2 + 80);
        j->nbb = orc_cc_decode_k7(j->soft, j->nsym + 1, dec);
        j->bits_b = NEW(uint8_t, j->nbb);
        orc_descramble(dec, j->nbb, 0x8A, 0x7F, 7, j->bits_b);
        free(dec); free(j->soft); j->soft = NULL;
        break;
    }
    }
}

static void* stage_thread(void* a)
{
    targ_t* t = (targ_t*)a;
    pipe_t* P = t->p;
    const int k = t->stage;
    for (;;) {
        job_t* j = q_get(&P->q[k]);
        if (!j) { q_put(&P->q[k + 1], NULL); break; }
        const double t0 = now_s();
        run_stage(P, k, j);
        P->busy[k] += now_s() - t0;
        q_put(&P->q[k + 1], j);
    }
    return NULL;
}

typedef struct { pipe_t* p; job_t* jobs; const cf32* iq; size_t n; int batch; } feed_t;
static void* feed_thread(void* a)
{
    feed_t* f = (feed_t*)a;
    for (int b = 0; b < f->batch; b++) { f->jobs[b].in = f->iq + (size_t)b * f->n; f->jobs[b].n = f->n; q_put(&f->p->q[0], &f->jobs[b]); }
    return NULL;
}

/* `batch` streams of n samples at 1 Msps through the thread-per-block pipeline of the 2FSK-1k receiver; returns the wall time, the
 * checksum orc_batch_rx(ORC_MODE_2FSK_1K, ...) returns for the same input, and (busy[0 .. 10]) the seconds every stage spent working */
double orc_pipeline_rx_2fsk1k(const cf32* iq, int batch, size_t n, double carrier_offset_hz, uint64_t* bit_checksum, double* busy, int* nstages)
{
    pipe_t P;
    memset(&P, 0, sizeof P);
    P.rot_inc = orc_phase_inc_to_turn(2 * M_PI * -carrier_offset_hz / 1000000.0);
    P.nrs = orc_low_pass(1, 1000000.0, 10000, 10000, ORC_WIN_BLACKMAN_HARRIS, NULL);
    P.rs = NEW(float, P.nrs); orc_low_pass(1, 1000000.0, 10000, 10000, ORC_WIN_BLACKMAN_HARRIS, P.rs);
    P.nft = orc_low_pass(1, 20000, 2000, 2000, ORC_WIN_BLACKMAN_HARRIS, NULL);
    P.ft = NEW(float, P.nft); orc_low_pass(1, 20000, 2000, 2000, ORC_WIN_BLACKMAN_HARRIS, P.ft);
    P.nb = orc_complex_band_pass(1, 20000, -2000, 0, 2000, ORC_WIN_BLACKMAN_HARRIS, NULL);
    P.up = NEW(cf32, P.nb); P.lo = NEW(cf32, P.nb);
    orc_complex_band_pass(1, 20000, -2000, 0, 2000, ORC_WIN_BLACKMAN_HARRIS, P.up);
    orc_complex_band_pass(1, 20000, 0, 2000, 2000, ORC_WIN_BLACKMAN_HARRIS, P.lo);
    P.nst = orc_low_pass(1.0, 20000, 2000, 2000, ORC_WIN_HAMMING, NULL);
    P.st = NEW(float, P.nst); orc_low_pass(1.0, 20000, 2000, 2000, ORC_WIN_HAMMING, P.st);
    for (int k = 0; k <= NSTAGE; k++) q_init(&P.q[k]);
    job_t* jobs = (job_t*)calloc((size_t)batch, sizeof(job_t));
    pthread_t th[NSTAGE];
    targ_t ta[NSTAGE];
    const double t0 = now_s();
    for (int k = 0; k < NSTAGE; k++) { ta[k].p = &P; ta[k].stage = k; pthread_create(&th[k], NULL, stage_thread, &ta[k]); }
    /* the source is a thread of its own too (it blocks when the first queue is full); this thread is the sink */
    feed_t fd = {&P, jobs, iq, n, batch};
    pthread_t feeder;
    pthread_create(&feeder, NULL, feed_thread, &fd);
    uint64_t total = 0;
    for (int got = 0; got < batch; got++) {
        job_t* j = q_get(&P.q[NSTAGE]);
        uint64_t h = 1469598103934665603ull;
        for (size_t i = 0; i < j->nba; i++) h = (h ^ j->bits_a[i]) * 1099511628211ull;
        for (size_t i = 0; i < j->nbb; i++) h = (h ^ j->bits_b[i]) * 1099511628211ull;
        total += h;
        free(j->bits_a); free(j->bits_b);
    }
    pthread_join(feeder, NULL);
    q_put(&P.q[0], NULL);
    for (int k = 0; k < NSTAGE; k++) pthread_join(th[k], NULL);
    const double dt = now_s() - t0;
    if (bit_checksum) *bit_checksum = total;
    if (busy) for (int k = 0; k < NSTAGE; k++) busy[k] = P.busy[k];
    if (nstages) *nstages = NSTAGE;
    free(P.rs); free(P.ft); free(P.up); free(P.lo); free(P.st); free(jobs);
    return dt;
}
