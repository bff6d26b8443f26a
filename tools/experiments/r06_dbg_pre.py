import sys, os
sys.path.insert(0, "/root/repo")
import torch
import qradiolink_amd as q
ctx = q.Context(0)
B, n = 1024, 262144
iq = torch.view_as_complex(torch.randn((B, n, 2), device="cuda") * 0.05)
dem = q.Demod(ctx, 18, batch=B, max_chunk=n, device_samp_rate=1000000)
for _ in range(4):
    dem.process_async(iq)
dem.sync()
dem.close(); ctx.close()
