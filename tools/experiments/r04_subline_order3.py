#!/usr/bin/env python3
"""Third look: host enqueue time per C4 step, fresh and after C1 (with its parity check), and what a gc / thread-count change does."""
import argparse, gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import qradiolink_amd as q
args = argparse.Namespace(steps=100, warmup=3, config="c1", batch=0, nsamp=0, pad=0, no_extra=True, overlap=False, no_overlap=False, free_tx=False,
                          no_grouped=False, fll_slim=False, cluster=False, no_marks=False, check=False, legacy_pfb=0, gpus=1)
dev = torch.device("cuda", 0)
ctx = q.Context(0)
import numpy as np
def c4(tag):
    B, n = 64, 1 << 21
    g = torch.Generator(device=dev); g.manual_seed(1)
    iq = torch.view_as_complex(torch.randn((B, n, 2), generator=g, device=dev) * 0.05)
    ch = q.Channelizer(ctx, 64, batch=B, max_chunk=n)
    ch.enable_4fsk()
    for _ in range(3): ch.process_async(iq)
    ch.sync()
    t0 = time.perf_counter(); host = []
    for _ in range(20):
        a = time.perf_counter(); ch.process_async(iq); host.append(time.perf_counter() - a)
    t1 = time.perf_counter(); ch.sync(); t2 = time.perf_counter()
    print("%-28s step %.3f ms   host enqueue per step: median %.3f ms max %.3f ms   threads %d" % (tag, (t2 - t0) / 20 * 1e3, sorted(host)[10] * 1e3, max(host) * 1e3, len(os.listdir("/proc/self/task"))), flush=True)
    ch.close(); del iq; torch.cuda.empty_cache()
c4("fresh")
r = bench.run_workload("c1", args, torch, q, ctx, dev, 0, 1, check=False)
c4("after c1 (no check)")
time.sleep(2); c4("  + 2 s")
r = bench.run_workload("c1", args, torch, q, ctx, dev, 0, 1, check=True, steps=20)
c4("after c1 with parity check")
time.sleep(2); c4("  + 2 s")
gc.collect(); c4("  + gc")
os.environ["OMP_WAIT_POLICY"] = "PASSIVE"
import orc
try:
    orc.lib.omp_set_num_threads
except Exception as e:
    pass
time.sleep(5); c4("  + 5 s")
