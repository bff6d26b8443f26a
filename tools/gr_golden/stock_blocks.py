#!/usr/bin/env python3
"""Block-level pins of the [GR-MEM] items that no hier-block run isolates (needs the GNU Radio 3.10 Python modules; NOT runnable in this repository's container):

    python tools/gr_golden/stock_blocks.py            # runs every case in GNU Radio, compares with the oracle, writes tests/golden/gr/stock_<name>.npy
    python tools/gr_golden/stock_blocks.py --dry-run  # no GNU Radio: lists the cases and evaluates the ORACLE side of each

Cases = the restatements docs/ORACLE_AND_PINS.md marks "from memory" and whose bits matter below the 1e-5 bound: the firdes designers the chains and their setters
call (window formulas, tap counts, normalisation), and analog::sig_source_f (fixed-point NCO + 1024-row sine table: the TX CTCSS tone and the CW key's source --
the least certain item of the oracle).  A case passes when the arrays are EQUAL (designs are deterministic double arithmetic narrowed to float; the NCO is integer
phase + float table arithmetic); a difference is reported with its first index and size, and the .npy of GNU Radio's output is written either way so that
tests/test_golden.py::test_oracle_against_stock_block_fixtures holds the oracle to it from then on."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))

BH, HAMMING = 5, 0   # gr::fft::window::win_type values (= the oracle's ORC_WIN_*)


def cases():
    """name -> (how GNU Radio produces it, how the oracle produces it); both return a numpy array"""
    import orc

    def gr_firdes(fn, *a):
        def run():
            from gnuradio import fft, filter as grf
            args = [fft.window.win_type(v) if isinstance(v, _Win) else v for v in a]
            t = getattr(grf.firdes, fn)(*args)
            return np.asarray(t, np.complex64 if fn.startswith("complex") else np.float32)
        return run

    def gr_sig(wave, freq, ampl, offset, n):
        def run():
            from gnuradio import analog, blocks, gr
            tb = gr.top_block()
            src = analog.sig_source_f(8000, getattr(analog, wave), freq, ampl, offset)
            head, snk = blocks.head(gr.sizeof_float, n), blocks.vector_sink_f()
            tb.connect(src, head, snk)
            tb.run()
            return np.asarray(snk.data(), np.float32)
        return run

    W = _Win
    return {
        # designers, with the arguments the chains and their setters use (tests/test_ref_chains.py has the construction logs)
        "low_pass_scope": (gr_firdes("low_pass", 1, 1000000, 50000, 25000, W(HAMMING)), lambda: orc.low_pass(1, 1000000, 50000, 25000)),
        "low_pass_frontend_1to50": (gr_firdes("low_pass", 1, 1000000, 10000, 10000, W(BH)), lambda: orc.low_pass(1, 1000000, 10000, 10000, orc.WIN_BH)),
        "low_pass_nbfm_setter": (gr_firdes("low_pass", 1, 20000, 4000, 1200, W(BH)), lambda: orc.low_pass(1, 20000, 4000, 1200, orc.WIN_BH)),
        "low_pass_2_nbfm_ctor": (gr_firdes("low_pass_2", 1, 20000, 5000, 3500, 60, W(BH)), lambda: orc.low_pass_2(1, 20000, 5000, 3500, 60, orc.WIN_BH)),
        "low_pass_2_dmr_interp": (gr_firdes("low_pass_2", 125, 3000000, 5000, 2000, 60, W(BH)), lambda: orc.low_pass_2(125, 3000000, 5000, 2000, 60, orc.WIN_BH)),
        "band_pass_2_ctcss_audio": (gr_firdes("band_pass_2", 1, 8000, 300, 3500, 200, 35, W(BH)), lambda: orc.band_pass_2(1, 8000, 300, 3500, 200, 35, orc.WIN_BH)),
        "band_pass_2_ssb_setter": (gr_firdes("band_pass_2", 2, 8000, 200, 2400, 200, 90, W(BH)), lambda: orc.band_pass_2(2, 8000, 200, 2400, 200, 90, orc.WIN_BH)),
        "complex_band_pass_am_setter": (gr_firdes("complex_band_pass", 1, 20000, -4000, 4000, 1200, W(BH)), lambda: orc.complex_band_pass(1, 20000, -4000, 4000, 1200, orc.WIN_BH)),
        "complex_band_pass_fll_edge": (gr_firdes("complex_band_pass", 1, 20000, 0, 2000, 2000, W(BH)), lambda: orc.complex_band_pass(1, 20000, 0, 2000, 2000, orc.WIN_BH)),
        "complex_band_pass_2_usb": (gr_firdes("complex_band_pass_2", 1, 8000, 200, 2700, 200, 90, W(BH)), lambda: orc.complex_band_pass_2(1, 8000, 200, 2700, 200, 90, orc.WIN_BH)),
        "rrc_dmr": (gr_firdes("root_raised_cosine", 5, 24000, 4800, 0.2, 125), lambda: orc.root_raised_cosine(5, 24000, 4800, 0.2, 125)),
        "rrc_qpsk": (gr_firdes("root_raised_cosine", 2, 2, 1, 0.35, 22), lambda: orc.root_raised_cosine(2, 2, 1, 0.35, 22)),
        # analog::sig_source_f: the TX CTCSS tone (gr_mod_nbfm.cpp:53) and the CW key's source (gr_mod_base.cpp:144)
        "sig_source_cos_88_5": (gr_sig("GR_COS_WAVE", 88.5, 0.15, 0, 16000), lambda: orc.sig_source_cos(8000, 88.5, 0.15, 16000)),
        "sig_source_cos_250_3": (gr_sig("GR_COS_WAVE", 250.3, 0.15, 0, 16000), lambda: orc.sig_source_cos(8000, 250.3, 0.15, 16000)),
        "sig_source_sin_600_key_down": (gr_sig("GR_SIN_WAVE", 600, 0.98, 1, 16000), lambda: orc.sig_source_sin(8000, 600, 0.98, 16000, offset=1.0)),
        "sig_source_sin_600_key_up": (gr_sig("GR_SIN_WAVE", 600, 0.001, 1, 16000), lambda: orc.sig_source_sin(8000, 600, 0.001, 16000, offset=1.0)),
    }


class _Win(int):
    """marks a window argument (turned into gr::fft::window::win_type on the GNU Radio side)"""


def compare(name, got, want):
    """-> None when equal, else a one-line description of the first difference"""
    got, want = np.asarray(got), np.asarray(want)
    if got.shape != want.shape:
        return "%s: %d items from GNU Radio, %d from the oracle" % (name, got.size, want.size)
    a, b = got.view(np.float32).astype(np.float64), want.view(np.float32).astype(np.float64)
    d = np.flatnonzero(got.view(np.uint32) != want.view(np.uint32))
    if d.size == 0:
        return None
    scale = max(float(np.sqrt(np.mean(b * b))), 1e-30)
    return "%s: %d of %d floats differ, first at %d (GNU Radio %.9g, oracle %.9g), max |diff| = %.3g = %.3g of RMS" % (
        name, d.size, a.size, d[0], a[d[0]], b[d[0]], float(np.max(np.abs(a - b))), float(np.max(np.abs(a - b))) / scale)


def dry_run(out=sys.stdout):
    """the oracle side of every case (no GNU Radio): name -> array"""
    res = {}
    for name, (_, oracle) in cases().items():
        res[name] = np.asarray(oracle())
        assert res[name].size > 0 and np.all(np.isfinite(res[name].view(np.float32))), name
        print("would run: %-32s oracle gives %d %s items" % (name, res[name].size, res[name].dtype), file=out)
    return res


def main():
    if "--dry-run" in sys.argv:
        dry_run()
        return 0
    try:
        import gnuradio  # noqa: F401
    except ImportError:
        raise SystemExit("the GNU Radio 3.10 Python modules are required (this image has none); --dry-run lists the cases")
    gr_dir = os.path.join(ROOT, "tests", "golden", "gr")
    os.makedirs(gr_dir, exist_ok=True)
    bad = 0
    for name, (gr_side, oracle) in cases().items():
        got, want = gr_side(), np.asarray(oracle())
        np.save(os.path.join(gr_dir, "stock_" + name + ".npy"), got)
        msg = compare(name, got, want)
        print("%-32s %s" % (name, "equal" if msg is None else "DIFFERS"))
        if msg:
            print("   " + msg)
            bad += 1
    print("%d of %d cases differ; tests/golden/gr/stock_*.npy written -- commit them" % (bad, len(cases())))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
