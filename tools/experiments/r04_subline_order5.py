#!/usr/bin/env python3
"""Fifth look: which part of the parity check flips the process into the state in which C4 takes 4.5 instead of 3.1 ms per step?"""
import argparse, gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import bench
import qradiolink_amd as q
dev = torch.device("cuda", 0)
ctx = q.Context(0)
def c4(tag):
    B, n = 64, 1 << 21
    g = torch.Generator(device=dev); g.manual_seed(1)
    iq = torch.view_as_complex(torch.randn((B, n, 2), generator=g, device=dev) * 0.05)
    ch = q.Channelizer(ctx, 64, batch=B, max_chunk=n)
    ch.enable_4fsk()
    for _ in range(3): ch.process_async(iq)
    ch.sync()
    t0 = time.perf_counter()
    for _ in range(20): ch.process_async(iq)
    ch.sync(); t2 = time.perf_counter()
    print("%-60s C4 step %.3f ms" % (tag, (t2 - t0) / 20 * 1e3), flush=True)
    ch.close(); del iq; torch.cuda.empty_cache()
c4("fresh")
x = torch.zeros(1 << 27, device=dev); y = x.cpu(); del x, y
c4("after a 512 MB device-to-host copy (pageable)")
import sig
iq = bench.synth("2fsk1k", 1000000, 1200.0, 16384, 262144, 1234, torch, dev, pad=0)
dem = q.Demod(ctx, 18, batch=16384, max_chunk=262144, device_samp_rate=1000000, carrier_offset_hz=1200.0, side_outputs=True)
for _ in range(5): dem.process_async(iq)
dem.sync()
c4("after 5 C1 calls (handle + 34 GB input alive)")
out = dem.process(iq)
c4("after a synchronous C1 call with its output tensors")
cnt = out["counts"].cpu().numpy()
c4("after counts.cpu()")
row = iq[7].cpu().numpy()
c4("after iq[7].cpu()")
ref = bench.oracle_demod("2fsk1k", row, 1000000, 1200.0)
c4("after the oracle on one stream (CPU)")
f = out["filtered"][7, :cnt[7, 0]].cpu().numpy()
c4("after filtered[7].cpu()")
dem.close(); del iq, out; torch.cuda.empty_cache()
c4("after closing the C1 handle and freeing its buffers")
