"""Times the rank-4 receivers (analogue voice, DSSS, M17) and the frame FEC kernels at batch sizes that fill the GPU: wall clock per
qrl_demod_process call (queued back to back, synchronised at the end), whole chain, synthetic input resident in HBM.
python tools/r02_extra_modes.py > profiles/r02_extra_modes.json"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import qradiolink_amd as q  # noqa: E402

ctx = q.Context(0)
res = {}
g = torch.Generator(device="cuda").manual_seed(1)
for name, modem, B, n in (("nbfm5000", q.MODEM_NBFM5000, 4096, 262144), ("am5000", q.MODEM_AM5000, 4096, 262144), ("usb2500", q.MODEM_USB2500, 4096, 262144),
                          ("wbfm", q.MODEM_WBFM, 4096, 262144), ("dsss_bpsk8", q.MODEM_BPSK8, 4096, 262144), ("m17", q.MODEM_M17, 4096, 262144)):
    iq = (torch.randn((B, n), device="cuda", generator=g) + 1j * torch.randn((B, n), device="cuda", generator=g)).to(torch.complex64) * 0.05
    dem = q.Demod(ctx, modem, batch=B, max_chunk=n)
    for _ in range(2):
        dem.process(iq)
    torch.cuda.synchronize()
    steps = 5
    t0 = time.perf_counter()
    for _ in range(steps):
        dem.process_async(iq)
    dem.sync()
    dt = (time.perf_counter() - t0) / steps
    dem.close()
    res[name] = {"streams": B, "samples_per_stream": n, "ms_per_call": round(dt * 1e3, 3), "GS_per_s": round(B * n / dt / 1e9, 2)}
    del iq
n = 1 << 20
bursts = torch.randint(0, 256, (n, 33), dtype=torch.uint8, device="cuda", generator=g)
frames = torch.randint(0, 256, (n, 48), dtype=torch.uint8, device="cuda", generator=g)
frames[:, 0] = 0xFF; frames[:, 1] = 0x5D
for name, fn, x in (("bptc19696_decode", q.bptc19696_decode, bursts), ("m17_decode_frames", q.m17_decode_frames, frames)):
    fn(ctx, x)
    t0 = time.perf_counter()
    for _ in range(3):
        fn(ctx, x)
    dt = (time.perf_counter() - t0) / 3
    res[name] = {"frames": n, "ms_per_call": round(dt * 1e3, 3), "Mframes_per_s": round(n / dt / 1e6, 1)}
print(json.dumps(res, indent=1))
