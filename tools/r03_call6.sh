#!/bin/bash
# round 3, GPU call 6: is the 5.06 TB/s wall of all three C1 front ends (VGPR phase-lane, LDS-DMA phase-lane, matrix-pipe) the DATA?
# the streaming microbenchmark on constant bytes vs pseudo-random floats, 16 and 32 GiB
set -u
export TMPDIR=/tmp
O=gpurun_out/r03f
rm -rf $O; mkdir -p $O
for cfg in "16 0" "16 1" "32 1" "32 0"; do
  timeout 120 ./build/stream_lds $cfg 1 >> $O/stream_lds_data.log 2>&1
done
cat $O/stream_lds_data.log
# the same question on the real kernel: the bench input replaced by zeros / constant
python - > $O/fe_data.log 2>&1 <<'P'
import sys, time, torch
sys.path.insert(0, '.')
import qradiolink_amd as q
ctx = q.Context(0)
B, n = 16384, 1 << 18
dem = q.Demod(ctx, 18, batch=B, max_chunk=n, device_samp_rate=1000000, carrier_offset_hz=1200.0, side_outputs=True)
for name in ("zeros", "const", "randn", "randn_small"):
    if name == "zeros": iq = torch.zeros((B, n), dtype=torch.complex64, device="cuda")
    elif name == "const": iq = torch.full((B, n), 0.01 + 0.02j, dtype=torch.complex64, device="cuda")
    else:
        g = torch.Generator(device="cuda"); g.manual_seed(1)
        iq = torch.empty((B, n), dtype=torch.complex64, device="cuda")
        v = torch.view_as_real(iq)
        for b0 in range(0, B, 1024): v[b0:b0 + 1024] = torch.randn((1024, n, 2), generator=g, device="cuda") * (0.05 if name == "randn" else 1e-6)
    for _ in range(2): dem.process_async(iq)
    dem.sync(); dem.profile(True)
    t0 = time.perf_counter()
    for _ in range(8): dem.process_async(iq)
    dem.sync(); dt = time.perf_counter() - t0
    kms, l, kn = dem.profile_read(); dem.profile(False)
    print("%-12s step %.3f ms  %s %.3f ms = %.0f GB/s" % (name, dt / 8 * 1e3, kn, kms / l, B * n * 8.178 / (kms / l * 1e-3) / 1e9), flush=True)
    del iq; torch.cuda.empty_cache()
P
cat $O/fe_data.log
