#!/usr/bin/env python3
"""The one-rank cluster step of C4 at the bench shape, one step at a time (a sync after every step): the kernels of the two handles alone on the chip,
timed with the handles' own HIP events.  Usage: python tools/c4_cluster_alone.py [steps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import qradiolink_amd as q
from qradiolink_amd import sharding
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ctx = q.Context(0)
B, n = 64, 1 << 21
g = torch.Generator(device="cuda"); g.manual_seed(1)
iq = torch.view_as_complex(torch.randn((B, n, 2), generator=g, device="cuda") * 0.05)
ex = sharding.Exchange.self_()
cl = sharding.Cluster(ctx, ex, 64, B, n)
cl.tail.enable_4fsk()
for _ in range(2):
    cl.step_async(iq); cl.sync()
cl.front.profile(True); cl.tail.profile(True)
for _ in range(steps):
    cl.step_async(iq); cl.sync()
print("cluster alone: front", "  ".join("%s %.3f ms" % (k, ms / max(c, 1)) for k, ms, c in cl.front.profile_read_kernels() if c),
      "| tail", "  ".join("%s %.3f ms" % (k, ms / max(c, 1)) for k, ms, c in cl.tail.profile_read_kernels() if c))
cl.close(); ex.close(); ctx.close()
