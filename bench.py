#!/usr/bin/env python3
"""bench.py — IQ MSamples/s through the RX demod chain on MI355X (BASELINE.json metric).

A "step" is one pass of the whole hot path (rotator + front-end decimator + per-mode resampler +
filters + symbol sync + 2x Viterbi + descramblers) over one batch of synthetic IQ that is already
resident in HBM.  Default workload = BASELINE.json configs[1]: GMSK 10 kbit/s RX chain on 25 Msps IQ, 384 streams x
1.6 M samples (65 ms of signal) per step = 5 GB of IQ per step.  (The serial symbol-sync tail costs ~0.4 us per symbol and stream:
with 96 streams x 6.5 M samples it was longer than the front end and capped the whole chain at 219 GS/s.)
The 2FSK-1k chain the north-star target is quoted on (configs[0], 1 Msps IQ) is measured too and
reported under "north_star_c1" in the same JSON line.

Launch: python bench.py --gpus N --steps K --warmup W   (N>1: via torch.distributed.run, one rank
per GPU; streams are sharded across ranks, there is no data-path collective => "scaling": "weak").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8 TB/s

WORKLOADS = {
    # name: (label, sig mode, modem type, device rate, rx offset, default batch, default samples/stream, oracle mode)
    "c2": ("C2: GMSK-10k RX chain (gr_demod_base front end 25:1 + gr_demod_gmsk) on 25 Msps IQ",
           "gmsk10k", 22, 25000000, 25000.0, 384, 25 * (1 << 16), 1),
    "c1": ("C1: 2FSK-1k RX chain (rotator + gr_demod_2fsk) on 1 Msps IQ",
           "2fsk1k", 18, 1000000, 1200.0, 16384, 1 << 18, 0),
    # not part of the default line (parity-test configs measured on request: --config c3 / c4)
    "c3": ("C3: QPSK-250k RX chain (gr_demod_base front end 100:1 + gr_demod_qpsk) on 100 Msps IQ",
           "qpsk250k", 26, 100000000, 25000.0, 384, 100 * (1 << 14), 2),
}


def synth(mode, device_rate, offset, batch, nsamp, seed, torch, dev):
    """Synthetic batch, built once (untimed): one oracle-modulated stream, then per-stream circular
    shift + CFO + AWGN applied on the GPU (torch is plumbing here, not the product)."""
    import sig
    nframes = {"gmsk10k": 6, "2fsk1k": 6, "qpsk250k": 4}[mode]
    base, _ = sig.make_stream(mode, nframes=nframes, device_rate=device_rate, rx_offset_hz=offset, seed=seed, amp=0.05)
    reps = -(-nsamp // base.size)
    base = np.tile(base, reps)[:nsamp]
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    x = torch.from_numpy(base).to(dev)
    iq = torch.empty((batch, nsamp), dtype=torch.complex64, device=dev)
    n = torch.arange(nsamp, device=dev, dtype=torch.int64)
    # a few large batched torch ops (groups of streams) instead of one small op per stream: rocprofv3 --pmc survives it
    group = max(1, min(batch, (1 << 27) // nsamp))
    for b0 in range(0, batch, group):
        bs = torch.arange(b0, min(b0 + group, batch), device=dev, dtype=torch.int64)
        shift = (7919 * bs) % nsamp
        cfo = 5.0 * ((bs % 21) - 10).to(torch.float64)
        idx = (n[None, :] - shift[:, None]) % nsamp
        ph = (cfo[:, None] / device_rate) * n[None, :].to(torch.float64)
        ph = (ph - torch.floor(ph)) * (2 * np.pi)
        rot = torch.polar(torch.ones_like(ph, dtype=torch.float32), ph.to(torch.float32))
        noise = torch.randn((bs.numel(), nsamp, 2), generator=g, device=dev, dtype=torch.float32) * 0.002
        iq[b0:b0 + bs.numel()] = x[idx] * rot + torch.view_as_complex(noise)
        del idx, ph, rot, noise
    return iq


def run_workload(name, args, torch, q, ctx, dev, rank, world, no_overlap=False):
    label, mode, modem, rate, offset, dbatch, dns, _ = WORKLOADS[name]
    if no_overlap:
        os.environ["QRL_NO_OVERLAP"] = "1"   # read by qrl_demod_create: stage C back on the main stream, kernels run one at a time
    else:
        os.environ.pop("QRL_NO_OVERLAP", None)
    batch = args.batch if (args.batch and name == args.config) else dbatch
    nsamp = args.nsamp if (args.nsamp and name == args.config) else dns
    nsamp &= ~1
    iq = synth(mode, rate, offset, batch, nsamp, 1234 + rank, torch, dev)
    dem = q.Demod(ctx, modem, batch=batch, max_chunk=nsamp, device_samp_rate=rate, carrier_offset_hz=offset,
                  side_outputs=True)
    for _ in range(args.warmup):
        dem.process_async(iq)
    dem.sync()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dem.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        dem.process_async(iq)
    dem.sync()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    kms, launches, kname = dem.profile_read()
    dem.profile(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    counts = dem.counts.cpu().numpy()
    dem.close()
    os.environ.pop("QRL_NO_OVERLAP", None)
    del iq
    torch.cuda.empty_cache()
    total_samples = float(batch) * nsamp * args.steps * world
    # dominant kernel roofline: algorithmic bytes per launch = input cf32 read once + decimated cf32 written once
    fe_decim = rate // 1000000 if rate >= 2000000 else 50
    bytes_per_launch = batch * nsamp * 8.0 * (1.0 + 1.0 / fe_decim)
    ach = bytes_per_launch / (kms / max(launches, 1) * 1e-3) / 1e9 if kms > 0 else 0.0
    return dict(name=name, default_shape=(batch == dbatch and nsamp == (dns & ~1)),
                label=label, batch=batch, nsamp=nsamp, rate=rate, seconds=dt, msps=total_samples / dt / 1e6,
                ms_per_step=dt / args.steps * 1e3, kernel=kname, kernel_ms=kms / max(launches, 1), launches=launches,
                achieved_gbps=ach, bits_per_stream=int(counts[:, 2].mean()), bytes_per_launch=bytes_per_launch)


def run_c4(args, torch, q, ctx, dev, world):
    """C4: multi-carrier MMDVM receiver, 64 x 25 kHz channels from 1.6 Msps wideband IQ (PFB channelizer + per-channel
    24/25 resampler, LPF, FM discriminator -> int16, RSSI tags and the 4FSK symbol tail).  Reported on request only."""
    M, B, n = 64, args.batch or 64, (args.nsamp or (1 << 21)) // 64 * 64
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    iq = torch.view_as_complex(torch.randn((B, n, 2), generator=g, device=dev, dtype=torch.float32) * 0.05)
    ch = q.Channelizer(ctx, M, batch=B, max_chunk=n)
    ch.enable_4fsk()
    for _ in range(args.warmup):
        ch.process_async(iq)
    ch.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ch.process_async(iq)
    ch.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ch.close()
    return {"metric": "wideband IQ MSamples/sec through the C4 receiver", "value": round(B * n * args.steps * world / dt / 1e6, 1), "unit": "MS/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C4: 64 x 25 kHz MMDVM channels from 1.6 Msps IQ: PFB channelizer + FM int16 + RSSI + 4FSK tail",
                       "wideband_streams_per_gpu": B, "samples_per_stream_per_step": n}}


def run_c5(args, torch, q, ctx, dev, world):
    """C5: full duplex -- QPSK-250k modulator and QPSK-250k demodulator handles on their own HIP streams, calls interleaved without
    synchronisation (BASELINE config 5; reference src/radiocontroller.cpp:2043-2078 runs the two top blocks concurrently)."""
    import sig
    B = args.batch or 4096
    n = (args.nsamp or (1 << 16)) & ~1
    nbytes = n // 32                                  # the TX produces as many 1 Msps samples as the RX consumes
    base, _ = sig.make_stream("qpsk250k", nframes=3, device_rate=1000000, seed=3, amp=0.05)
    base = np.tile(base, -(-n // base.size))[:n]
    iq = torch.from_numpy(base).to(dev).repeat(B, 1).contiguous()
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    data = torch.randint(0, 256, (B, nbytes), generator=g, device=dev, dtype=torch.uint8)
    dem = q.Demod(ctx, 26, batch=B, max_chunk=n)
    mod = q.Mod(ctx, 26, batch=B, max_bytes=nbytes)
    tx_out = torch.empty((B, nbytes * mod.spb), dtype=torch.complex64, device=dev)

    def loop(k, do_tx=True, do_rx=True):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            if do_tx:
                mod.process_async(data, out=tx_out)
            if do_rx:
                dem.process_async(iq)
        mod.sync(); dem.sync()
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    loop(args.warmup)
    dt = loop(args.steps)
    dt_rx = loop(args.steps, do_tx=False)
    dt_tx = loop(args.steps, do_rx=False)
    dem.close(); mod.close()
    tot = float(B) * n * args.steps * world
    return {"metric": "IQ MSamples/sec through RX demod chain (with the TX chain running concurrently)", "value": round(tot / dt / 1e6, 1),
            "unit": "MS/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C5: full duplex QPSK-250k TX + RX at 1 Msps on two HIP streams", "streams_per_gpu": B,
                       "samples_per_stream_per_step": n, "tx_msps_concurrent": round(tot / dt / 1e6, 1),
                       "rx_alone_ms_per_step": round(dt_rx / args.steps * 1e3, 3), "tx_alone_ms_per_step": round(dt_tx / args.steps * 1e3, 3)}}


def cpu_baseline(name, threads, budget_s=12.0):
    """Oracle (CPU restatement of the reference flowgraph) on a bounded sample of the same workload:
    `threads` independent streams (OpenMP over streams, one stream per core), repeated until ~budget_s seconds
    of CPU wall time have been measured."""
    import orc
    import sig
    label, mode, modem, rate, offset, _, _, omode = WORKLOADS[name]
    nstreams = max(threads, 1)
    per = (1 << 22) if rate >= 2000000 else (1 << 20)
    base, _ = sig.make_stream(mode, nframes=2, device_rate=rate, rx_offset_hz=offset, seed=99, amp=0.05)
    base = np.tile(base, -(-per // base.size))[:per]
    iq = np.stack([np.roll(base, 977 * b) for b in range(nstreams)]).astype(np.complex64)
    total, reps = 0.0, 0
    while total < budget_s and reps < 200:
        secs, _ = orc.batch_rx(omode, iq, rate, offset, threads)
        total += secs
        reps += 1
    return dict(value=round(reps * nstreams * per / total / 1e6, 3), unit="MS/s", cores=threads, kind="port",
                sample="%d passes over %d streams x %d samples of the %s workload (%.1f s of CPU wall time), "
                       "oracle/liborc.so (C, -O3, OpenMP over streams)" % (reps, nstreams, per, name.upper(), total))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c2", choices=sorted(WORKLOADS) + ["c4", "c5"])
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--nsamp", type=int, default=0)
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary workload and the CPU baseline")
    ap.add_argument("--no-overlap", action="store_true", help="developer aid: run the kernels of a call one after another")
    args = ap.parse_args()

    import torch
    import qradiolink_amd as q

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl", device_id=dev)
    ctx = q.Context(local)

    if args.config in ("c4", "c5"):
        line = (run_c4 if args.config == "c4" else run_c5)(args, torch, q, ctx, dev, world)
        if rank == 0:
            print(json.dumps(line))
        ctx.close()
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    main_r = run_workload(args.config, args, torch, q, ctx, dev, rank, world, no_overlap=args.no_overlap)
    extra = None
    base = None
    if not args.no_extra and args.config in ("c1", "c2"):
        other = "c1" if args.config == "c2" else "c2"
        extra = run_workload(other, args, torch, q, ctx, dev, rank, world)
        if rank == 0:
            base = cpu_baseline(args.config, min(os.cpu_count() or 1, 16))
    # C1 (2FSK family) runs in overlapped mode: the FLL / discriminator kernels of call k share the GPU with the front end of
    # call k + 1, which stretches the front-end kernel.  Its stand-alone duration is measured in a second short pass.
    alone = {}
    for r in (main_r, extra):
        if r and r["name"] == "c1" and not args.no_extra:
            alone["c1"] = run_workload("c1", args, torch, q, ctx, dev, rank, world, no_overlap=True)
    if rank == 0:
        def roof(r):
            # HBM traffic per launch of the dominant kernel: PMC numbers cannot be collected from inside this process;
            # they come from the separate rocprofv3 --pmc passes of tools/gpu_profile_final.sh (FETCH_SIZE doubled per
            # the gfx950 correction + WRITE_SIZE), stored in profiles/pmc_traffic.json for the DEFAULT workload shape.
            traffic = None
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(r["name"])
                if pmc and r["default_shape"] and r["kernel"] in pmc["kernel"]:
                    traffic = pmc["fetch_bytes"] + pmc["write_bytes"]
            except (OSError, ValueError, KeyError):
                pass
            d = dict(bound="hbm", achieved=round(r["achieved_gbps"], 1), peak=HBM_PEAK_GBPS, unit="GB/s",
                     frac=round(r["achieved_gbps"] / HBM_PEAK_GBPS, 4), traffic=traffic, kernel=r["kernel"],
                     kernel_ms=round(r["kernel_ms"], 4), launches=r["launches"],
                     algorithmic_bytes_per_launch=r["bytes_per_launch"])
            a = alone.get(r["name"])
            if a:
                d["without_overlap"] = dict(kernel_ms=round(a["kernel_ms"], 4), achieved=round(a["achieved_gbps"], 1),
                                            frac=round(a["achieved_gbps"] / HBM_PEAK_GBPS, 4), ms_per_step=round(a["ms_per_step"], 3),
                                            note="QRL_NO_OVERLAP=1: same workload with the kernels of a call run one after another")
            return d
        line = {
            "metric": "IQ MSamples/sec through RX demod chain", "value": round(main_r["msps"], 1), "unit": "MS/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(main_r["ms_per_step"], 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": main_r["label"], "streams_per_gpu": main_r["batch"],
                       "samples_per_stream_per_step": main_r["nsamp"], "device_samp_rate": main_r["rate"],
                       "parallelism": "streams sharded over ranks, no collective",
                       "decoded_bits_per_stream_last_step": main_r["bits_per_stream"]},
            "roofline": roof(main_r),
        }
        if base:
            line["cpu_baseline"] = base
        if extra:
            key = "north_star_c1" if args.config == "c2" else "c2"
            line[key] = {"workload": extra["label"], "value": round(extra["msps"], 1), "unit": "MS/s",
                         "ms_per_step": round(extra["ms_per_step"], 3), "streams_per_gpu": extra["batch"],
                         "samples_per_stream_per_step": extra["nsamp"], "roofline": roof(extra)}
        print(json.dumps(line))
    ctx.close()
    if world > 1:
        torch.distributed.barrier()   # rank 0 is behind by the CPU baseline: leave together
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
