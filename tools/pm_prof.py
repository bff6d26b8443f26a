#!/usr/bin/env python3
"""Phase profile of k_decim_pm (build/libqrl_pmprof.so, -DQRL_PM_PROF): QRL_LIB_PATH=build/libqrl_pmprof.so python tools/pm_prof.py [c1|c2|c3]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import qradiolink_amd as q
ctx = q.Context(0)
CFG = {"c1": (18, 1000000, 1200.0, 16384, 1 << 18), "c2": (22, 25000000, 25000.0, 384, 25 * (1 << 16)), "c3": (26, 100000000, 25000.0, 384, 100 * (1 << 14))}
modem, rate, offset, B, n = CFG[sys.argv[1] if len(sys.argv) > 1 else "c1"]
CH = min(B, 64)
g = torch.Generator(device="cuda"); g.manual_seed(1)
iq = torch.empty((B, n), dtype=torch.complex64, device="cuda")
v = torch.view_as_real(iq)
for b0 in range(0, B, CH):
    v[b0:b0 + CH] = torch.randn((CH, n, 2), generator=g, device="cuda") * 0.05
dem = q.Demod(ctx, modem, batch=B, max_chunk=n, device_samp_rate=rate, carrier_offset_hz=offset, side_outputs=True)
lib = ctx.lib
out = (C.c_ulonglong * 8)()
for _ in range(2): dem.process_async(iq)
dem.sync(); lib.qrl_pm_prof_read(out)
for _ in range(4): dem.process_async(iq)
dem.sync(); lib.qrl_pm_prof_read(out)
names = ["wait vmcnt (group landed)", "raw LDS reads + lgkmcnt(0)", "DMA issue", "rotate + MFMA", "diag sums + store", "", "", "groups"]
ng = out[7]
tot = sum(out[k] for k in range(5))
for k in range(5):
    print("%-32s %8.0f cycles per group and wave  (%4.1f %%)" % (names[k], out[k] / ng, 100.0 * out[k] / tot))
print("total %.0f cycles per group and wave; groups per call %.0f" % (tot / ng, ng / 4))
