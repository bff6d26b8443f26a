#!/usr/bin/env python3
"""Phase profile of k_fec (developer build -DQRL_FEC_PROF, tools/kernel_variants.sh kernels_fec.hip fecprof -DQRL_FEC_PROF):
QRL_LIB_PATH=build/libqrl_fecprof.so python tools/fec_prof.py  -- shader-clock ticks per phase, wave and 80-bit block (C5 RX shape, a
QPSK-250k signal so that the trellis sees what the bench gives it)."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import qradiolink_amd as q
import sig
ctx = q.Context(0)
B, n = 16384, 16384
base, _ = sig.make_stream("qpsk250k", nframes=3, device_rate=1000000, seed=3, amp=0.05)
base = np.tile(base, -(-n // base.size))[:n]
iq = torch.from_numpy(base).cuda().repeat(B, 1).contiguous()
dem = q.Demod(ctx, 26, batch=B, max_chunk=n)
lib = ctx.lib
out = (C.c_ulonglong * 8)()
for _ in range(2): dem.process_async(iq); dem.sync()
lib.qrl_fec_prof_read(out)
for _ in range(4): dem.process_async(iq); dem.sync()
lib.qrl_fec_prof_read(out)
names = ["symbol selects + setup", "(unused)", "forward (86 steps, table pre-passes included)", "history complement", "end state + chainback (scalar, 80 steps x 2)", "descrambler + stores"]
nb = out[7]
tot = sum(out[k] for k in range(6))
for k in range(6):
    print("%-36s %8.0f ticks per block and wave  (%4.1f %%)" % (names[k], out[k] / nb, 100.0 * out[k] / tot))
print("total %.0f ticks per block and wave; blocks x waves per call %.0f" % (tot / nb, nb / 4))
