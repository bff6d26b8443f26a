#!/usr/bin/env python3
"""Is the slow mode of the C1 front end (8.3 instead of 6.9 ms, decided per process) a property of the process's memory placement?
Times a plain torch reduction over the same 34 GB input next to the front-end kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import qradiolink_amd as q
ctx = q.Context(0)
B, n = 16384, 262144
iq = torch.empty((B, n), dtype=torch.complex64, device="cuda")
iq.real.normal_(0, 0.05); iq.imag.normal_(0, 0.05)
v = torch.view_as_real(iq).view(-1)
def t_sum():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): s = v.sum()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 5
t_sum()
ts = t_sum()
dem = q.Demod(ctx, 18, batch=B, max_chunk=n, carrier_offset_hz=1200.0)
for _ in range(3): dem.process_async(iq)
dem.sync(); dem.profile(True)
for _ in range(20): dem.process_async(iq)
dem.sync()
kms, launches, name = dem.profile_read()
print("torch sum: %.2f ms = %.2f TB/s   %s: %.3f ms   data_ptr %% 2^30 = %#x" % (ts * 1e3, v.numel() * 4 / ts / 1e12, name, kms / launches, iq.data_ptr() % (1 << 30)))
