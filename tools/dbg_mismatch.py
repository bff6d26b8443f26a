import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import orc, sig, qradiolink_amd as q
ctx = q.Context(0)
rate, offset = int(os.environ.get("RATE", "25000000")), 25000.0
B = 8
iq = sig.make_batch("gmsk10k", B, nframes=2, device_rate=rate, rx_offset_hz=offset, seed=3)
refs = [orc.demod_gmsk(orc.frontend(iq[b], rate, offset), sps=1, filter_width=20000)["filtered"].view(np.uint32).reshape(-1, 2) for b in range(B)]
d = torch.from_numpy(iq).cuda()
nbad = 0
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for rep in range(reps):
    dem = q.Demod(ctx, 22, batch=B, max_chunk=1 << 23, device_samp_rate=rate, carrier_offset_hz=offset)
    out = q.collect(dem, d, 1 << 23)
    dem.close()
    for b in range(B):
        g = out["filtered"][b].view(np.uint32).reshape(-1, 2)
        bad = np.nonzero((g != refs[b]).any(axis=1))[0]
        if bad.size:
            nbad += 1
            print("rep", rep, "stream", b, "bad", bad.size, "first", bad[0], "tile", int(bad[0] * 12.5 / 256),
                  "got", out["filtered"][b][bad[0]], "want", refs[b][bad[0]].view(np.float32), hex(g[bad[0]][0]), hex(refs[b][bad[0]][0]))
print("env", {k: v for k, v in os.environ.items() if k.startswith("QRL_")}, "bad stream-runs:", nbad, "of", reps * B)
