#!/usr/bin/env python3
"""Second look at the slow sub-lines of the default bench run: replay its sequence with / without the parity checks and pauses."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import qradiolink_amd as q
args = argparse.Namespace(steps=100, warmup=3, config="c1", batch=0, nsamp=0, pad=0, no_extra=True, overlap=False, no_overlap=False, free_tx=False,
                          no_grouped=False, fll_slim=False, cluster=False, no_marks=False, check=False, legacy_pfb=0, gpus=1)
dev = torch.device("cuda", 0)
ctx = q.Context(0)
def c4(tag, **kw):
    r = bench.run_c4(args, torch, q, ctx, dev, 0, 1, steps=20, with_form2=kw.get("form2", False), check=kw.get("check", False))
    print(tag, "c4 ms/step", r["ms_per_step"], "median", r["step_spread_ms"]["median"], flush=True)
mode = sys.argv[1] if len(sys.argv) > 1 else "a"
r = bench.run_workload("c1", args, torch, q, ctx, dev, 0, 1, check=(mode != "nocheck"))
print("c1", round(r["ms_per_step"], 3), flush=True)
r = bench.run_workload("c1", args, torch, q, ctx, dev, 0, 1, overlap=False, steps=20)
time.sleep(2)
r = bench.run_workload("c2", args, torch, q, ctx, dev, 0, 1, steps=50, check=(mode != "nocheck"))
print("c2", round(r["ms_per_step"], 3), flush=True)
time.sleep(2)
r = bench.run_workload("c3", args, torch, q, ctx, dev, 0, 1, steps=30, check=(mode != "nocheck"))
print("c3", round(r["ms_per_step"], 3), flush=True)
time.sleep(2)
c4("default-like", form2=True, check=(mode != "nocheck"))
time.sleep(2)
c4("again, no form 2 / check")
time.sleep(2)
c4("again with form 2", form2=True)
