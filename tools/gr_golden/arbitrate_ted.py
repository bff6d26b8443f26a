#!/usr/bin/env python3
"""arbitrate_ted.py <gr_golden output dir> <iq dir>: which modified-M&M formula does the installed GNU Radio implement?

Port 1 of every reference demodulator IS the symbol_sync output (gr_demod_2fsk.cpp:145, gr_demod_gmsk.cpp:112,
gr_demod_qpsk.cpp:141 connect the loop's output to the constellation port), so the dump of gr_golden already carries the
signal that localises the one [GR-MEM] formula the sensitivity record (tests/golden/ted_sensitivity.json) shows moving hard
bits.  For each dumped chain the oracle is run on the same IQ under each candidate of include/qrl_contracts.h and the first
symbol index where the real port 1 leaves each candidate is printed; the candidate that stays within 1e-5 of RMS is upstream's.
Needs the outputs of run_all.py (a GNU Radio 3.10 machine); nothing to do in this repository's container."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import orc           # noqa: E402
import make_golden   # noqa: E402

NAMES = {0: "clip(u/2,1)", 1: "clip(u,1)/2", 2: "clip(u,1)"}


def arbitrate(out_dir, iq_dir, only=None, out=sys.stdout):
    """-> {(case, candidate): first symbol beyond 1e-5 of RMS, or None}"""
    res = {}
    for name, mode, rate, offset, (kind, kw) in make_golden.CASES:
        f1 = os.path.join(out_dir, name + ".port1")
        if rate != 1000000 or not os.path.exists(f1) or kind not in ("2fsk", "gmsk", "qpsk", "4fsk") or (only and name not in only):
            continue
        real = np.fromfile(f1, np.complex64)
        x = np.fromfile(os.path.join(iq_dir, name + ".cf32"), np.complex64)
        which = "cc" if kind == "qpsk" or (kind == "4fsk" and not kw.get("fm", False)) else "ff"
        for v in (0, 1, 2):
            orc.lib.orc_set_ted_modmm(v if which == "ff" else -1, v if which == "cc" else -1)
            try:
                got = getattr(orc, "demod_" + kind)(x, **kw)["constellation"]
            finally:
                orc.lib.orc_set_ted_modmm(-1, -1)
            n = min(got.size, real.size)
            rms = np.sqrt(np.mean(np.abs(real[:n]) ** 2)) + 1e-30
            off = np.nonzero(np.abs(got[:n] - real[:n]) / rms > 1e-5)[0]
            res[(name, v)] = int(off[0]) if off.size else None
            print("%-14s symbol_sync_%s %-12s first symbol beyond 1e-5: %s of %d" % (name, which, NAMES[v], off[0] if off.size else "none", n), file=out)
    return res


def self_test(cases=("2fsk1k_1M", "qpsk250k_1M"), out=sys.stdout):
    """Dry run without GNU Radio: the "real" port 1 is minted by the oracle under its own contract from the committed fixture inputs; the
    contract's candidate must then stay within 1e-5 on every symbol and another candidate must leave it -- proves the tool's plumbing
    (case table, IQ export format, candidate switch, comparison) end to end.  Returns the result dictionary."""
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        for name, mode, rate, offset, (kind, kw) in make_golden.CASES:
            if name not in cases:
                continue
            z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
            x = z["iq_f16"].astype(np.float32).view(np.complex64)                      # what make_golden.py --export-iq writes
            x.tofile(os.path.join(d, name + ".cf32"))
            getattr(orc, "demod_" + kind)(x, **kw)["constellation"].astype(np.complex64).tofile(os.path.join(d, name + ".port1"))
        return arbitrate(d, d, only=cases, out=out)


if __name__ == "__main__":
    if len(sys.argv) == 2 and sys.argv[1] == "--self-test":
        sys.exit(0 if self_test() else 1)
    if len(sys.argv) != 3:
        raise SystemExit(__doc__)
    arbitrate(sys.argv[1], sys.argv[2])
