/* oracle/orc_trace.c — TEST INFRASTRUCTURE (part of the CPU oracle; never linked into the product).
 *
 * A construction trace of the oracle's chains: while enabled, every block-level primitive appends one line "name(args)" when a chain
 * calls it, and every filter designer registers the taps it produced under the text of the design call, so that a FIR primitive
 * prints "fir_ccf(low_pass(1,40000,4000,4000,5))" rather than numbers.  tests/test_ref_chains.py compares this trace with the
 * construction log of the reference's own hier-block constructors (oracle/ref_shim_rec.cpp): same blocks, same parameters, same
 * order of processing.  Off by default and then free; single-threaded (enable it around one chain call on one thread).
 * floats print as %.9g, doubles as %.17g — both round-trip, so equal text <=> equal value. */
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "orc.h"

static int g_on = 0;
/* block timing (bench.py, cpu_baseline "thread_per_block_model"): with the trace on, the time between two primitive calls of a chain
 * is the run time of the first one (the chains call their blocks one after the other over the whole stream) */
#define ORC_MAX_TIMED 256
static int g_time_on = 0, g_nblk = 0;
static double g_tprev = 0.0, g_secs[ORC_MAX_TIMED];
static char g_bname[ORC_MAX_TIMED][40];
static double clock_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
static char* g_buf = NULL;
static size_t g_len = 0, g_cap = 0;
typedef struct { uint64_t h; char* text; } design_t;
static design_t* g_des = NULL;
static size_t g_ndes = 0, g_capdes = 0;

static uint64_t fnv(const void* p, size_t n)
{
    uint64_t h = 1469598103934665603ull;
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ull;
    return h ^ n;
}

void orc_trace_enable(int on)
{
    g_on = on;
    g_len = 0;
    if (g_buf) g_buf[0] = 0;
    for (size_t i = 0; i < g_ndes; i++) free(g_des[i].text);
    g_ndes = 0;
}

int orc_trace_on(void) { return g_on; }

const char* orc_trace_get(void) { return g_buf ? g_buf : ""; }

static void append(const char* s)
{
    size_t n = strlen(s);
    if (g_len + n + 2 > g_cap) { g_cap = (g_len + n + 2) * 2; g_buf = (char*)realloc(g_buf, g_cap); }
    memcpy(g_buf + g_len, s, n); g_len += n;
    g_buf[g_len++] = '\n'; g_buf[g_len] = 0;
}

void orc_trace_event(const char* fmt, ...)
{
    if (!g_on) return;
    char line[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(line, sizeof line, fmt, ap); va_end(ap);
    append(line);
    if (g_time_on) {
        const double t = clock_s();
        if (g_nblk > 0) g_secs[g_nblk - 1] += t - g_tprev;
        if (g_nblk < ORC_MAX_TIMED) {
            size_t k = 0;
            while (line[k] && line[k] != '(' && k + 1 < sizeof g_bname[0]) { g_bname[g_nblk][k] = line[k]; k++; }
            g_bname[g_nblk][k] = 0;
            g_secs[g_nblk++] = 0.0;
        }
        g_tprev = clock_s();
    }
}

/* on: clear and start (also turns the construction trace on); off: close the last block */
void orc_block_timing_enable(int on)
{
    if (on) {   /* what runs before the first traced primitive is gr_demod_base's rotator_cc (orc_frontend) */
        orc_trace_enable(1); g_nblk = 1; g_secs[0] = 0.0; strcpy(g_bname[0], "rotator_cc"); g_time_on = 1; g_tprev = clock_s();
    }
    else if (g_time_on) { if (g_nblk > 0) g_secs[g_nblk - 1] += clock_s() - g_tprev; g_time_on = 0; g_on = 0; }
}
int orc_block_timing_count(void) { return g_nblk; }
double orc_block_timing_get(int i, char* name, size_t cap)
{
    if (i < 0 || i >= g_nblk) return -1.0;
    if (name && cap) { strncpy(name, g_bname[i], cap - 1); name[cap - 1] = 0; }
    return g_secs[i];
}

void orc_trace_taps(const void* taps, size_t bytes, const char* fmt, ...)
{
    if (!g_on || !taps) return;
    char line[512];
    va_list ap; va_start(ap, fmt); vsnprintf(line, sizeof line, fmt, ap); va_end(ap);
    if (g_ndes == g_capdes) { g_capdes = g_capdes ? g_capdes * 2 : 16; g_des = (design_t*)realloc(g_des, g_capdes * sizeof *g_des); }
    g_des[g_ndes].h = fnv(taps, bytes);
    g_des[g_ndes].text = strdup(line);
    g_ndes++;
}

/* the design call that produced these taps, or "taps[bytes]" when they were not designed under the trace */
const char* orc_trace_name(const void* taps, size_t bytes)
{
    static char anon[64];
    if (!g_on) return "";
    uint64_t h = fnv(taps, bytes);
    for (size_t i = g_ndes; i-- > 0;) if (g_des[i].h == h) return g_des[i].text;
    snprintf(anon, sizeof anon, "taps[%zu]", bytes);
    return anon;
}
