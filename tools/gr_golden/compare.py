#!/usr/bin/env python3
"""compare.py <gr_golden output dir> <tests/golden dir>: real-reference ports against the oracle-minted fixtures.
Bits must match exactly (after the flowgraph's start-up the streams are aligned by construction: same block histories);
float ports are compared by SHA only when identical, else by max |difference| / RMS against the 1e-5 of the north star."""
import glob
import os
import sys

import numpy as np

out_dir, gold_dir = sys.argv[1], sys.argv[2]
bad = 0
for path in sorted(glob.glob(os.path.join(gold_dir, "*.npz"))):
    name = os.path.basename(path)[:-4]
    z = np.load(path)
    for port, key, nkey in ((2, "bits_a", "n_bits_a"), (3, "bits_b", "n_bits_b")):
        f = os.path.join(out_dir, "%s.port%d" % (name, port))
        if not os.path.exists(f) or int(z[nkey]) == 0:
            continue
        got = np.fromfile(f, np.uint8)
        want = np.unpackbits(z[key])[: int(z[nkey])]
        n = min(got.size, want.size)
        ok = n > 0 and np.array_equal(got[:n], want[:n])
        print("%-16s port %d: %d / %d bits %s" % (name, port, n, want.size, "equal" if ok else "DIFFER"))
        bad += not ok
    f = os.path.join(out_dir, name + ".port0")
    if os.path.exists(f):
        got = np.fromfile(f, np.complex64)[:64]
        want = z["filtered_head"]
        n = min(got.size, want.size)
        rms = np.sqrt(np.mean(np.abs(want[:n]) ** 2)) + 1e-30
        err = np.max(np.abs(got[:n] - want[:n])) / rms
        print("%-16s port 0 head: max |diff| / rms = %.3g %s" % (name, err, "ok" if err <= 1e-5 else "ABOVE 1e-5"))
        bad += err > 1e-5
sys.exit(1 if bad else 0)
