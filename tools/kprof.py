#!/usr/bin/env python3
"""Kernel times of ONE call at a time (sync after every call: no pipelining across calls, so every kernel runs alone):
    rocprofv3 --kernel-trace --stats -d out -o x -- python tools/kprof.py <modem> <batch> <nsamp> [device_rate] [calls]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import qradiolink_amd as q

modem, batch, nsamp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rate = int(sys.argv[4]) if len(sys.argv) > 4 else 1000000
calls = int(sys.argv[5]) if len(sys.argv) > 5 else 4
ctx = q.Context(0)
g = torch.Generator(device="cuda")
g.manual_seed(1)
iq = torch.view_as_complex(torch.randn((batch, nsamp, 2), generator=g, device="cuda") * 0.05)
dem = q.Demod(ctx, modem, batch=batch, max_chunk=nsamp, device_samp_rate=rate)
if os.environ.get("QRL_KPROF_UNFUSED"):      # A/B: the 1:2 resampler and the RRC of the QPSK chain as two kernels
    dem.set_option(q.OPT_UNFUSED_DEC2, 1)
if os.environ.get("QRL_KPROF_GROUPED"):
    dem.set_option(q.OPT_GROUPED, int(os.environ["QRL_KPROF_GROUPED"]))
for _ in range(calls):
    dem.process_async(iq)
    if not os.environ.get("QRL_KPROF_PIPELINED"):   # QRL_KPROF_PIPELINED=1: the calls queued back to back, as the bench does (tools/prof_timeline.py shows how they overlap)
        dem.sync()
dem.sync()
dem.close()
ctx.close()
