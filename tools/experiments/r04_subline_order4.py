#!/usr/bin/env python3
"""Fourth look: is it WHERE the buffers land?  One C4 handle + input created at the start and reused, against ones created after C1 + parity check."""
import argparse, gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import qradiolink_amd as q
args = argparse.Namespace(steps=100, warmup=3, config="c1", batch=0, nsamp=0, pad=0, no_extra=True, overlap=False, no_overlap=False, free_tx=False,
                          no_grouped=False, fll_slim=False, cluster=False, no_marks=False, check=False, legacy_pfb=0, gpus=1)
dev = torch.device("cuda", 0)
ctx = q.Context(0)
B, n = 64, 1 << 21
def make():
    g = torch.Generator(device=dev); g.manual_seed(1)
    iq = torch.view_as_complex(torch.randn((B, n, 2), generator=g, device=dev) * 0.05)
    ch = q.Channelizer(ctx, 64, batch=B, max_chunk=n)
    ch.enable_4fsk()
    return iq, ch
def run(tag, iq, ch):
    for _ in range(3): ch.process_async(iq)
    ch.sync()
    t0 = time.perf_counter()
    for _ in range(20): ch.process_async(iq)
    ch.sync(); t2 = time.perf_counter()
    print("%-44s step %.3f ms   iq at %#x" % (tag, (t2 - t0) / 20 * 1e3, iq.data_ptr()), flush=True)
iq0, ch0 = make()
run("old handle, fresh", iq0, ch0)
r = bench.run_workload("c1", args, torch, q, ctx, dev, 0, 1, check=True, steps=20)
run("old handle, after c1 + parity check", iq0, ch0)
iq1, ch1 = make()
run("new handle + input, after c1 + parity check", iq1, ch1)
run("old input, new handle", iq0, ch1)
run("new input, old handle", iq1, ch0)
print(torch.cuda.memory_summary(abbreviated=True)[:600])
