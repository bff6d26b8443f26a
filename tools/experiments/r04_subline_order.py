#!/usr/bin/env python3
"""Why are the c2-c5 sub-lines of the default bench run slower than the same workloads run alone?  One process: C4 (20 steps) before C1,
right after C1, after C1 + a pause, after C1 + torch.cuda.empty_cache()."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import qradiolink_amd as q
args = argparse.Namespace(steps=100, warmup=3, config="c1", batch=0, nsamp=0, pad=0, no_extra=True, overlap=False, no_overlap=False, free_tx=False,
                          no_grouped=False, fll_slim=False, cluster=False, no_marks=False, check=False, legacy_pfb=0, gpus=1)
dev = torch.device("cuda", 0)
ctx = q.Context(0)
def c4(tag):
    r = bench.run_c4(args, torch, q, ctx, dev, 0, 1, steps=20, with_form2=False)
    print(tag, "c4 ms/step", r["ms_per_step"], "median", r["step_spread_ms"]["median"], "mem GB", round(torch.cuda.memory_reserved() / 1e9, 1), flush=True)
c4("fresh       ")
r = bench.run_workload("c1", args, torch, q, ctx, dev, 0, 1)
print("c1 ms/step", round(r["ms_per_step"], 3), flush=True)
c4("after c1    ")
time.sleep(2.0)
c4("after pause ")
torch.cuda.empty_cache()
c4("after empty ")
r = bench.run_workload("c2", args, torch, q, ctx, dev, 0, 1, steps=50)
print("c2 ms/step", round(r["ms_per_step"], 3), flush=True)
c4("after c2    ")
