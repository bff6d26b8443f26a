#!/usr/bin/env python3
"""Which stage of k_qpsk_pipe4 sets the pace (developer build -DQRL_Q4_PROF, tools/kernel_variants.sh kernels_qpsk.hip q4prof -DQRL_Q4_PROF):
QRL_LIB_PATH=build/libqrl_q4prof.so python tools/q4_prof.py [batch] [nsamp] -- per wave the share of the loop it spent working (the rest is
waiting at the window barrier for the slowest stage)."""
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 384
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
ctx = q.Context(0)
base, _ = sig.make_stream("qpsk250k", nframes=3, device_rate=1000000, seed=3, amp=0.05)
base = np.tile(base, -(-n // base.size))[:n]
iq = torch.from_numpy(base).cuda().repeat(B, 1).contiguous()
dem = q.Demod(ctx, 26, batch=B, max_chunk=n)
out = (C.c_ulonglong * 18)()
for _ in range(2): dem.process_async(iq); dem.sync()
ctx.lib.qrl_q4_prof_read(out)
for _ in range(4): dem.process_async(iq); dem.sync()
ctx.lib.qrl_q4_prof_read(out)
names = ["wave 0  agc2_cc", "wave 1  costas_loop_cc (1st)", "wave 2  symbol_sync_cc", "wave 3  costas (2nd) + diff_phasor", "wave 4  load next window / flush", "wave 5  load next window / flush"]
for w in range(6):
    busy, total, cnt = out[3 * w], out[3 * w + 1], out[3 * w + 2]
    print("%-36s busy %9.0f of %9.0f ticks per workgroup and call = %5.1f %%" % (names[w], busy / cnt, total / cnt, 100.0 * busy / total))
print("samples per stream and call at 500 ksps: %d -> %.0f ticks per sample for the loop" % (n // 2, out[1] / out[2] / (n // 2)))
