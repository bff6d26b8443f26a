/*
 * qrl_contracts.h — NAMED arithmetic contracts for stock GNU Radio block formulas that this tree restates from memory
 * ([GR-MEM]) and that no file under /root/reference can pin.  One definition, used by
 *   - the HIP kernels (qradiolink_amd/csrc/kernels_loops.hip, kernels_qpsk.hip)   : compile-time selection,
 *   - the CPU oracle  (oracle/orc_blocks.c)                                       : the same default, plus a run-time override
 *     (orc_set_ted_modmm) that tests/test_ted_sensitivity.py uses to measure what each candidate formula would change.
 * Nothing here is test infrastructure: it is the statement of what the product computes.
 *
 * ---- modified Mueller & Muller timing error detector -------------------------------------------------------------------
 * Reference call sites (all pass gr::digital::TED_MOD_MUELLER_AND_MULLER to symbol_sync_ff / symbol_sync_cc):
 *   /root/reference/src/gr/gr_demod_2fsk.cpp:108, gr_demod_gmsk.cpp:90, gr_demod_qpsk.cpp:107, gr_demod_4fsk.cpp:135-137,
 *   gr_demod_m17.cpp:72.
 * Upstream (gr-digital/lib/timing_error_detector.cc, GNU Radio 3.10, as recalled):
 *   ted_mod_mueller_and_muller::compute_error_ff():  u = (x0 - x2) d1 - (d0 - d2) x1;            return branchless_clip(u / 2.0f, 1.0f);
 *   ted_mod_mueller_and_muller::compute_error_cf():  u = (x0 - x2) conj(d1) - (d0 - d2) conj(x1); return branchless_clip(u.real(), 1.0f);
 * i.e. the real-valued body halves BEFORE the clip and the complex body does not halve at all.  SURVEY.md A.6 wrote the
 * real-valued form as clip(u, 1) / 2; the three candidates differ only where |u| > 1 (acquisition, 4-level symbols).
 */
#ifndef QRL_CONTRACTS_H
#define QRL_CONTRACTS_H

#define QRL_TED_MODMM_HALVE_BEFORE_CLIP 0   /* e = clip(u / 2, 1) */
#define QRL_TED_MODMM_HALVE_AFTER_CLIP  1   /* e = clip(u, 1) / 2 */
#define QRL_TED_MODMM_NONE              2   /* e = clip(u, 1)     */

/* the contract the product implements (and the oracle's default) */
#define QRL_TED_MODMM_FF QRL_TED_MODMM_HALVE_BEFORE_CLIP   /* symbol_sync_ff: C1, C2, 4FSK (FM), M17 */
#define QRL_TED_MODMM_CC QRL_TED_MODMM_NONE                /* symbol_sync_cc: C3 / C5 (QPSK), 4FSK discriminator branch */

/* CLIP(x, limit) is the caller's branchless_clip (device or host flavour: same arithmetic, gnuradio/math.h) */
#define QRL_TED_MODMM_ERROR(variant, u, CLIP)                                         \
    ((variant) == QRL_TED_MODMM_HALVE_BEFORE_CLIP ? CLIP((u) / 2.0f, 1.0f)            \
     : (variant) == QRL_TED_MODMM_HALVE_AFTER_CLIP ? CLIP((u), 1.0f) / 2.0f           \
                                                   : CLIP((u), 1.0f))

#endif
