#!/usr/bin/env python3
"""Phase profile of k_chan_tail (developer build -DQRL_CT_PROF, tools/kernel_variants.sh kernels_chan_tail.hip):
QRL_LIB_PATH=build/libqrl_<name>.so python tools/ct_prof.py  -- shader-clock ticks per phase, wave and tile (C4 bench shape)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import qradiolink_amd as q
ctx = q.Context(0)
B, n = 64, 1 << 21
g = torch.Generator(device="cuda"); g.manual_seed(1)
iq = torch.view_as_complex(torch.randn((B, n, 2), generator=g, device="cuda") * 0.05)
ch = q.Channelizer(ctx, 64, batch=B, max_chunk=n)
ch.enable_4fsk()
lib = ctx.lib
out = (C.c_ulonglong * 16)()
for _ in range(2): ch.process_async(iq)
ch.sync(); lib.qrl_ct_prof_read(out)
for _ in range(4):
    ch.process_async(iq)
    if os.environ.get("QRL_CT_PROF_SYNC"):   # every call alone on the chip (no channelizer / symbol synchroniser of a neighbouring call beside the kernel)
        ch.sync()
ch.sync(); lib.qrl_ct_prof_read(out)
names = ["tables issue", "input loads issue", "input -> LDS (waits for the loads)", "barrier", "A resampler", "barrier", "B channel filter", "barrier",
         "D discriminators + int16", "barrier", "(all waves) E / C", "waves 0-2: E RRC (same ticks as previous row, split)", "wave 3: C RSSI sums"]
nw = out[15]
tot = sum(out[k] for k in range(11))
for k in range(11):
    print("%-52s %8.0f cycles per tile and wave  (%4.1f %%)" % (names[k], out[k] / nw, 100.0 * out[k] / tot))
print("%-52s %8.0f per wave of waves 0-2;  wave 3 (RSSI): %8.0f" % ("E / C split", out[11] / (nw * 0.75), out[12] / (nw * 0.25)))
print("total %.0f cycles per tile and wave; wave-tiles per call %.0f" % (tot / nw, nw / 4))
