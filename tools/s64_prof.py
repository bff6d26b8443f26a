#!/usr/bin/env python3
"""Phase profile of k_pfb_stream64 (developer build -DQRL_S64_PROF, tools/kernel_variants.sh kernels_chan.hip):
QRL_LIB_PATH=build/libqrl_<name>.so python tools/s64_prof.py [streams [samples]]   -- shader-clock ticks per phase, wave and tile."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import qradiolink_amd as q
ctx = q.Context(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 21
g = torch.Generator(device="cuda"); g.manual_seed(1)
iq = torch.view_as_complex(torch.randn((B, n, 2), generator=g, device="cuda") * 0.05)
ch = q.Channelizer(ctx, 64, batch=B, max_chunk=n)
lib = ctx.lib
out = (C.c_ulonglong * 8)()
for _ in range(2): ch.process_async(iq)
ch.sync(); lib.qrl_s64_prof_read(out)
for _ in range(4): ch.process_async(iq)
ch.sync(); lib.qrl_s64_prof_read(out)
names = ["fetch issue (tile ahead)", "branch FIRs (VALU + LDS)", "barrier 1", "bin 32 + matrix phase", "s_waitcnt vmcnt (pieces landed)", "DPP + stores", "barrier 2"]
nt = out[7]
tot = sum(out[k] for k in range(7))
for k in range(7):
    print("%-34s %8.0f cycles per tile and wave  (%4.1f %%)" % (names[k], out[k] / nt, 100.0 * out[k] / tot))
print("total %.0f cycles per tile and wave; wave-tiles per call %.0f" % (tot / nt, nt / 4))
