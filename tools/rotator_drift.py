#!/usr/bin/env python3
"""How far VOLK's rotator recursion (volk_32fc_s32fc_x2_rotator_32fc generic: phase *= phase_inc in complex float, renormalised every
512 samples) is from the ideal phasor, per carrier offset: the first sample at which the deviation exceeds 1e-6 / 1e-5 / 1e-4, and
the deviation after 2^20 samples.  The kernels' rotator is the exact 2^-64-turn NCO (<= 5e-7 from the ideal at every sample,
tests/test_oracle.py::test_rotator_is_exact_nco), so this is also the distance between the two.  docs/ORACLE_AND_PINS.md quotes it."""
import numpy as np
import sys

def volk_phase(theta, n):
    inc = np.complex64(np.cos(theta) + 1j * np.sin(theta))
    ph = np.complex64(1.0)
    out = np.empty(n, np.complex64)
    for i in range(n):
        out[i] = ph
        ph = np.complex64(ph * inc)                      # complex float multiply: 4 products, 2 sums, each rounded to float
        if (i + 1) % 512 == 0:
            ph = np.complex64(ph / np.float32(np.hypot(np.float32(ph.real), np.float32(ph.imag))))
    return out

n = 1 << 20
for rate, hz in ((1e6, 1200.0), (1e6, 25000.0), (25e6, 1200.0), (100e6, 300000.0), (1.6e6, 12500.0)):
    theta = -2 * np.pi * hz / rate
    v = volk_phase(theta, n)
    ideal = np.exp(1j * theta * np.arange(n, dtype=np.float64))
    dev = np.abs(v.astype(np.complex128) - ideal)
    first = lambda t: int(np.argmax(dev > t)) if (dev > t).any() else -1
    print("fs %9.0f  offset %8.0f Hz: first sample beyond 1e-6 / 1e-5 / 1e-4 = %7d / %7d / %7d   deviation after 2^20 samples %.2e" %
          (rate, hz, first(1e-6), first(1e-5), first(1e-4), dev[-1]))
