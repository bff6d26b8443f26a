import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import orc, sig, qradiolink_amd as q
ctx = q.Context(0)
iq = sig.make_batch("2fsk1k", 2, nframes=2, device_rate=1000000, seed=3)
ref = orc.demod_2fsk(orc.frontend(iq[0], 1000000, 0.0), sps=10, filter_width=2000, fm=False)
for chunk in (1 << 21, 50000):
    dem = q.Demod(ctx, 18, batch=2, max_chunk=chunk, device_samp_rate=1000000, carrier_offset_hz=0.0)
    out = q.collect(dem, torch.from_numpy(iq).cuda(), chunk)
    dem.close()
    g, w = out["filtered"][0].view(np.uint32).reshape(-1, 2), ref["filtered"].view(np.uint32).reshape(-1, 2)
    bad = np.nonzero((g != w).any(axis=1))[0]
    print("chunk", chunk, "n", g.shape[0], w.shape[0], "bad", bad.size, bad[:10], "max rel", float(np.max(np.abs(out["filtered"][0] - ref["filtered"])) / np.max(np.abs(ref["filtered"]))))
