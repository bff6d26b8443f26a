#!/usr/bin/env python3
"""C4 at the bench shape, one call at a time (qrl_chan_sync after every call): every kernel ALONE on the chip, timed with the handle's own
HIP events (qrl_chan_profile_read_kernels).  QRL_LIB_PATH selects a variant library.  Usage: python tools/c4_alone.py [calls]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import qradiolink_amd as q
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ctx = q.Context(0)
B, n = 64, 1 << 21
g = torch.Generator(device="cuda"); g.manual_seed(1)
iq = torch.view_as_complex(torch.randn((B, n, 2), generator=g, device="cuda") * 0.05)
ch = q.Channelizer(ctx, 64, batch=B, max_chunk=n)
ch.enable_4fsk()
for _ in range(2):
    ch.process_async(iq); ch.sync()
ch.profile(True)
for _ in range(calls):
    ch.process_async(iq); ch.sync()
print("alone:", "  ".join("%s %.3f ms" % (k, ms / max(c, 1)) for k, ms, c in ch.profile_read_kernels()))
ch.close(); ctx.close()
