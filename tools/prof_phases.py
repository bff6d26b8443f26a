import ctypes as C, os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["QRL_DBG"] = "32"
import torch, qradiolink_amd as q
ctx = q.Context(0)
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
if cfg == "c2":
    B, N, rate, modem = 384, 25 * (1 << 16), 25000000, 22
else:
    B, N, rate, modem = 16384, 1 << 18, 1000000, 18
iq = torch.randn((B, N, 2), device="cuda").mul_(0.05)
iq = torch.view_as_complex(iq)
dem = q.Demod(ctx, modem, batch=B, max_chunk=N, device_samp_rate=rate, carrier_offset_hz=25000.0)
lib = q.load_library()
out = (C.c_ulonglong * 8)()
dem.process_async(iq); dem.sync()
lib.qrl_debug_decim_prof(out)
dem.process_async(iq); dem.sync()
lib.qrl_debug_decim_prof(out)
v = list(out)
n = max(v[7], 1)
names = ["t_hi + barrier", "wait loads", "commit", "barrier", "issue next loads", "mfma quarter (+alias barrier)", "barrier + combine + store"]
if os.environ.get("QRL_DECIM_TWO_TEAM") == "1":
    names = ["stage: wait loads", "stage: commit", "stage: issue+table", "mfma role", "combine", "barrier after mfma", "barrier after stage"]
tot = sum(v[:7])
print("workgroups", v[7], "ticks/WG", tot / n)
for k in range(7):
    print("%-14s %10.0f ticks/WG  %5.1f%%" % (names[k], v[k] / n, 100.0 * v[k] / tot))
