#!/usr/bin/env python3
"""bench.py — IQ MSamples/s through the RX demod chain on MI355X (BASELINE.json metric).

A "step" is one pass of the whole hot path (rotator + decimators + per-mode filters + carrier / timing recovery + Viterbi +
descramblers) over one batch of synthetic IQ that is already resident in HBM.

Default workload = C1, the configuration BASELINE.json's target is quoted on: 2FSK 1 kbit/s RX chain (gr_demod_2fsk behind the
gr_demod_base rotator) on 1 Msps IQ, 16384 streams x 262144 samples = 34.4 GB of IQ per step.  The same JSON line carries
  roofline      the HBM-facing kernel (k_decim_pl: rotator + 1:50 decimator) timed with HIP events on the handle's stream;
                achieved = SURVEY.md 8(d) algorithmic bytes (C1 8.178 B, C2 8.033 B, C3 8.0625 B per input sample) x samples
                per launch / average launch duration
  cpu_baseline  oracle/liborc.so timed on the host cores (a bounded sample of the same workload)
  parity_check  four random streams of one call AT THE BENCH SHAPE compared with the oracle (bits and port 0), untimed
  c2            the GMSK-10k chain on 25 Msps IQ (BASELINE configs[1]) measured in the same run.
Other configs on request: --config c2 | c3 | c4 | c5.

Launch: python bench.py --gpus N --steps K --warmup W.  N > 1 re-executes itself under torch.distributed.run (one rank per GPU,
RCCL); the driver may also launch it that way directly.  C1-C3/C5 shard independent streams over the ranks with no data-path
collective ("scaling": "weak"); C4 shards the CHANNELS: every rank channelizes its wideband streams and one RCCL all_to_all_single
per step hands each channel's samples to its owner ("scaling": "strong").
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
# what "bit-exact" in every parity_check of this file means: the checker is oracle/, not GNU Radio
PARITY_AGAINST = ("oracle/ (C restatement of the GNU Radio 3.10 block semantics; its FIR / DFT summation orders are contracts shared with the kernels, "
                  "each with a float64 definition test); the arithmetic of the stock GNU Radio / VOLK blocks is unpinned (no GNU Radio in this image), and the rotator "
                  "is an exact 2^-64-turn NCO that leaves VOLK's phase recursion by up to 4.5e-4 on the float ports at a 25 kHz offset (hard bits unaffected); "
                  "[GR-MEM] the fast_atan2f, tanhf_lut and MMSE-interpolator tables are regenerated from their formulas and rounded to float -- the literal upstream tables "
                  "are not available here")

# the channel every synth() stream of C1 / C2 / C3 / C5 goes through (VERDICT r5 "next" #1: parity_check names it)
SYNTH_CHANNEL = ("SURVEY 8(d) (tests/sig.py SPEC): fractional delay 0.37 sample + clock error +20 ppm (32-tap windowed-sinc resampling of the modulator output), "
                 "CFO, AWGN at Es/N0 12 dB per channel symbol; then per stream a circular shift, a CFO of -50 .. +50 Hz and AWGN of 0.002 rms per component")

WORKLOADS = {
    # name: (label, sig mode, modem type, device rate, rx offset, default batch, default samples/stream, oracle mode,
    #        algorithmic bytes per input sample: SURVEY.md 8(d))
    "c1": ("C1: 2FSK-1k RX chain (rotator + gr_demod_2fsk) on 1 Msps IQ",
           "2fsk1k", 18, 1000000, 1200.0, 16384, 1 << 18, 0, 8.178),
    "c2": ("C2: GMSK-10k RX chain (gr_demod_base front end 25:1 + gr_demod_gmsk) on 25 Msps IQ",
           "gmsk10k", 22, 25000000, 25000.0, 384, 25 * (1 << 16), 1, 8.033),
    "c3": ("C3: QPSK-250k RX chain (gr_demod_base front end 100:1 + gr_demod_qpsk) on 100 Msps IQ",
           "qpsk250k", 26, 100000000, 25000.0, 384, 100 * (1 << 14), 2, 8.0625),
}


def synth(mode, device_rate, offset, batch, nsamp, seed, torch, dev, pad=0):
    """Synthetic batch, built once (untimed): one oracle-modulated stream, then per-stream circular
    shift + CFO + AWGN applied on the GPU (torch is plumbing here, not the product)."""
    import sig
    nframes = {"gmsk10k": 6, "2fsk1k": 6, "qpsk250k": 4}[mode]
    # the base stream goes through SURVEY 8(d)'s channel (tests/sig.py SPEC: 0.37-sample fractional delay, +20 ppm clock error, Es/N0 12 dB)
    base, _ = sig.make_stream(mode, nframes=nframes, device_rate=device_rate, rx_offset_hz=offset, seed=seed, amp=0.05, impair=sig.SPEC)
    reps = -(-nsamp // base.size)
    base = np.tile(base, reps)[:nsamp]
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    x = torch.from_numpy(base).to(dev)
    iq = torch.empty((batch, nsamp + pad), dtype=torch.complex64, device=dev)[:, :nsamp]   # row pitch nsamp + pad samples
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


def oracle_demod(mode, x, rate, offset):
    import orc
    fe = orc.frontend(x, rate, offset)
    if mode == "2fsk1k":
        return orc.demod_2fsk(fe, sps=10, filter_width=2000, fm=False)
    if mode == "gmsk10k":
        return orc.demod_gmsk(fe, sps=1, filter_width=20000)
    return orc.demod_qpsk(fe, sps=2, filter_width=160000)


def parity_check(dem, iq, mode, rate, offset, torch, nstreams=4, seed=5):
    """One call from a fresh state at the bench shape; `nstreams` random streams against the oracle (untimed)."""
    dem.reset()
    out = dem.process(iq)
    cnt = out["counts"].cpu().numpy()
    rng = np.random.default_rng(seed)
    picks = sorted(int(b) for b in rng.choice(iq.shape[0], size=min(nstreams, iq.shape[0]), replace=False))
    two_branch = mode in ("2fsk1k", "gmsk10k")
    for b in picks:
        ref = oracle_demod(mode, iq[b].cpu().numpy(), rate, offset)
        got_a = out["bits_a"][b, :cnt[b, 2]].cpu().numpy()
        ok = np.array_equal(got_a, ref["bits_a"])
        if two_branch:
            ok = ok and np.array_equal(out["bits_b"][b, :cnt[b, 3]].cpu().numpy(), ref["bits_b"])
        got_f = out["filtered"][b, :cnt[b, 0]].cpu().numpy().view(np.float32) + np.float32(0)
        want_f = ref["filtered"].view(np.float32) + np.float32(0)
        ok = ok and got_f.size == want_f.size and np.array_equal(got_f.view(np.uint32), want_f.view(np.uint32))
        if not ok:
            return dict(status="FAILED", stream=b, streams=picks)
    return dict(status="bit-exact", streams=picks, compared="bits A/B and port 0 (filtered) of one call from a fresh state", channel=SYNTH_CHANNEL, against=PARITY_AGAINST,
                bits_per_stream=int(cnt[picks[0], 2]))


class StepMarks:
    """Per-step completion times that order NOTHING: after every queued step one event is recorded on each of the streams the step's
    kernels run on (qrl_*_internal_streams); a step is complete when the last of its events has fired, the differences of consecutive
    completion times are the intervals.  (Round 4 first took the marks on a side stream behind qrl_*_stream_wait.  Streams of one
    priority share a few hardware queues, and when the side stream landed on the queue of one of the handle's streams its wait -- for
    the whole step -- sat in front of the next step's kernel there: the steps ran strictly one after the other, C4 at 4.5 instead of
    3.05 ms.  Whether that happened depended on how many streams the process had created before: it hit the sub-lines of the default
    run, not the stand-alone runs.  tools/experiments/r04_subline_order*.py)"""

    def __init__(self, torch, streams, enabled=True, note="intervals between step completions (the last of the events recorded, per step, on the handle's internal streams)"):
        self.torch, self.enabled, self.note, self.ev = torch, enabled, note, []
        self.streams = [torch.cuda.ExternalStream(s) for s in streams] if enabled else []

    def mark(self):
        if not self.enabled:
            return
        evs = []
        for s in self.streams:
            e = self.torch.cuda.Event(enable_timing=True)
            e.record(s)
            evs.append(e)
        self.ev.append(evs)

    def spread(self):
        """min / p10 / median / p90 / max of the step intervals in ms (the first interval starts at the mark before the first timed step)"""
        if len(self.ev) < 2:
            return None
        base = self.ev[0][0]
        done = [max(base.elapsed_time(e) for e in evs) for evs in self.ev]
        series = [b - a for a, b in zip(done[:-1], done[1:])]
        d = sorted(series)
        if os.environ.get("QRL_BENCH_DUMP_STEPS"):
            sys.stderr.write("step intervals [ms]: " + " ".join("%.2f" % x for x in series) + "\n")
        return dict(min=round(d[0], 3), p10=round(d[len(d) // 10], 3), median=round(d[len(d) // 2], 3), p90=round(d[(9 * len(d)) // 10], 3), max=round(d[-1], 3), steps=len(d),
                    note=self.note)


def run_workload(name, args, torch, q, ctx, dev, rank, world, overlap=None, check=False, steps=None):
    label, mode, modem, rate, offset, dbatch, dns, _, abytes = WORKLOADS[name]
    steps = steps or args.steps
    batch = args.batch if (args.batch and name == args.config) else dbatch
    nsamp = args.nsamp if (args.nsamp and name == args.config) else dns
    nsamp &= ~1
    iq = synth(mode, rate, offset, batch, nsamp, 1234 + rank, torch, dev, pad=args.pad)
    dem = q.Demod(ctx, modem, batch=batch, max_chunk=nsamp, device_samp_rate=rate, carrier_offset_hz=offset,
                  side_outputs=True)
    if overlap is not None and name == "c1":
        dem.set_option(q.OPT_OVERLAP, 1 if overlap else 0)   # library default for the 2FSK family: 1 (decimated-rate kernels of call k under the front end of call k + 1)
    if getattr(args, "fll_slim", False):
        dem.set_option(q.OPT_FLL_SLIM, 1)
    for _ in range(args.warmup):
        dem.process_async(iq)
    dem.sync()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dem.profile(True)
    marks = StepMarks(torch, dem.internal_streams, enabled=not args.no_marks)
    marks.mark()
    t0 = time.perf_counter()
    for _ in range(steps):
        dem.process_async(iq)
        marks.mark()
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
    parity = parity_check(dem, iq, mode, rate, offset, torch) if (check and rank == 0) else None
    dem.close()
    del iq
    torch.cuda.empty_cache()
    total_samples = float(batch) * nsamp * steps * world
    bytes_per_launch = batch * nsamp * abytes          # per-sample figure x samples one launch processes
    ach = bytes_per_launch / (kms / max(launches, 1) * 1e-3) / 1e9 if kms > 0 else 0.0
    return dict(name=name, default_shape=(batch == dbatch and nsamp == (dns & ~1)), steps=steps,
                label=label, batch=batch, nsamp=nsamp, rate=rate, seconds=dt, msps=total_samples / dt / 1e6,
                ms_per_step=dt / steps * 1e3, kernel=kname, kernel_ms=kms / max(launches, 1), launches=launches,
                achieved_gbps=ach, bits_per_stream=int(counts[:, 2].mean()), bytes_per_launch=bytes_per_launch,
                bytes_per_sample=abytes, parity=parity, spread=marks.spread())


def timed_loop(fn, sync, args, torch, dev, world, marks=None):
    """W warm-up + K timed steps, barrier + synchronize on both sides, MAX over ranks."""
    for _ in range(args.warmup):
        fn()
    sync()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    if marks:
        marks.mark()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fn()
        if marks:
            marks.mark()
    sync()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    return dt


FRONT_END_FLOP = {"c1": 4.0 * 419 / 50 + 6.0, "c2": 4.0 * 1045 / 25 + 6.0, "c3": 4.0 * 4181 / 100 + 6.0}   # tap counts: docs/PATH_AND_BOUNDARY.md (low_pass of the 1:D stages)
C4_BYTES = 8.0 + 64 * 24000 * 2 / 1.6e6 + 64 * 4800 * 2 / 1.6e6   # input cf32 + int16 FM samples + unpacked dibits, per wideband sample
F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: f32 vector (packed) = f32 matrix peak
# algorithmic flop of the C4 contract per WIDEBAND sample (a real x complex MAC = 4 flop, a real MAC = 2):
#   channelizer: 35-tap branch FIR 140 + the 64-point DFT as the contract's direct sum with the conjugate-mirrored twiddle table 256
#   per 24 ksps channel sample (x 64 channels x 24000 / 1.6e6 = 0.96 per wideband sample): 24/25 resampler 35 taps 140, channel filter 33 taps
#   132, RRC 125 real taps 250, two discriminators + quantiser + |f|^4 sums ~ 40
C4_FLOP_PFB = 140.0 + 256.0
C4_FLOP_TAIL = (140.0 + 132.0 + 250.0 + 40.0) * 64 * 24000 / 1.6e6


def source_id():
    """Identifies the kernel sources a measurement belongs to: sha256 over qradiolink_amd/csrc + include (sorted by name)."""
    import hashlib
    h = hashlib.sha256()
    for d in ("qradiolink_amd/csrc", "include"):
        for fn in sorted(os.listdir(os.path.join(ROOT, d))):
            if fn.endswith((".hip", ".cpp", ".hpp", ".h")):
                h.update(fn.encode())
                with open(os.path.join(ROOT, d, fn), "rb") as f:
                    h.update(f.read())
    return h.hexdigest()[:12]


def pmc_traffic(name, kernel, default_shape=True):
    """HBM bytes per launch of the dominant kernel of workload `name`.  PMC counters cannot be read from inside this process; the
    number is the one measured by the separate rocprofv3 --pmc passes of tools/profile_round.sh on this same command (FETCH_SIZE with
    the calibrated gfx950 factor + WRITE_SIZE), kept in profiles/pmc_traffic.json per workload shape TOGETHER WITH the id of the
    kernel sources it was taken on: a different source id (the kernels changed since the pass) gives traffic = null."""
    try:
        allp = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        pmc = allp.get(name)
        if not pmc or not default_shape or kernel.split("<")[0].split(" ")[0] not in pmc["kernel"]:
            return None, None
        if allp.get("_source_id") != source_id():
            return None, "profiles/pmc_traffic.json was taken on kernel sources %s, these are %s: not reported" % (allp.get("_source_id"), source_id())
        return pmc["fetch_bytes"] + pmc["write_bytes"], pmc.get("source")
    except (OSError, ValueError, KeyError):
        return None, None


def pmc_chain_traffic(name, default_shape=True):
    """C4: HBM bytes per step of the WHOLE chain (sum over its kernels) and the per-kernel split, from the same PMC passes; (None, None) when
    the record is missing or belongs to other kernel sources."""
    try:
        allp = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        pmc = allp.get(name)
        if not pmc or not default_shape or "chain_bytes" not in pmc or allp.get("_source_id") != source_id():
            return None, None
        return pmc["chain_bytes"], {k: round(v) for k, v in pmc["chain_by_kernel"].items()}
    except (OSError, ValueError, KeyError):
        return None, None


# `traffic` is not measured by the run that prints it
TRAFFIC_FROM = "a separate rocprofv3 --pmc pass on these kernel sources (source id %s), committed under profiles/; NOT collected in this run"


def whole_step_obj(bytes_per_step, ms_per_step):
    """The same algorithmic bytes over the WHOLE step (every kernel of the chain, as timed by the bench loop) instead of the dominant
    kernel's own duration: printed beside `frac` on every line (hygiene)."""
    ach = bytes_per_step / (ms_per_step * 1e-3) / 1e9 if ms_per_step > 0 else 0.0
    return dict(achieved=round(ach, 1), frac=round(ach / HBM_PEAK_GBPS, 4), ms_per_step=round(ms_per_step, 3),
                note="algorithmic bytes of one step / ms_per_step / 8 TB/s: the whole chain, not the dominant kernel alone")


def roofline_obj(kernel, kms, launches, bytes_per_launch, bytes_per_sample, note=None, name=None, default_shape=True, ms_per_step=None):
    ach = bytes_per_launch / (kms / max(launches, 1) * 1e-3) / 1e9 if kms > 0 else 0.0
    traffic, src = pmc_traffic(name, kernel, default_shape) if name else (None, None)
    d = dict(bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBPS, unit="GB/s", frac=round(ach / HBM_PEAK_GBPS, 4), traffic=traffic,
             traffic_source=src, traffic_from=TRAFFIC_FROM % source_id() if traffic else None,
             kernel=kernel, kernel_ms=round(kms / max(launches, 1), 4), launches=launches,
             algorithmic_bytes_per_launch=bytes_per_launch, algorithmic_bytes_per_sample=bytes_per_sample)
    if ms_per_step:
        d["whole_step"] = whole_step_obj(bytes_per_launch, ms_per_step)
    if note:
        d["note"] = note
    return d


VALU_ISSUE_PEAK_G = 256 * 4 * 2.4 / 4      # 1024 SIMDs, one wave64 VALU instruction per 4 cycles each, 2.4 GHz (MI355X_MICROARCH.md): 614.4 G wave-instr/s


def issue_roofline(name, rx_seconds_per_call, default_shape=True):
    """The issue-side bound of a receiver call (C3, C5: the recursive per-stream chain makes the call VALU-issue / latency bound, not
    HBM bound): wave instructions of one RX call, counted by the SQ_INSTS_* PMC pass of tools/profile_round.sh on this same command
    (profiles/pmc_traffic.json, valid for the kernel sources it was taken on), over the RX-alone time of a call measured here,
    against 1024 SIMDs x one wave64 VALU instruction per 4 cycles."""
    try:
        allp = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        rec = allp.get(name + "_issue")
        if not rec or not default_shape:
            return None
        if allp.get("_source_id") != source_id():
            return dict(bound="issue", achieved=None, peak=VALU_ISSUE_PEAK_G, unit="G wave-instr/s", frac=None,
                        note="profiles/pmc_traffic.json was taken on kernel sources %s, these are %s: not reported" % (allp.get("_source_id"), source_id()))
        pc = rec["per_rx_call"]
        valu = pc.get("SQ_INSTS_VALU", 0.0)
        every = sum(pc.get(k, 0.0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH"))
        ach = valu / rx_seconds_per_call / 1e9
        return dict(bound="issue", achieved=round(ach, 1), peak=VALU_ISSUE_PEAK_G, unit="G wave-instr/s", frac=round(ach / VALU_ISSUE_PEAK_G, 4),
                    valu_wave_instr_per_rx_call=valu, all_wave_instr_per_rx_call=every, waves_per_rx_call=pc.get("SQ_WAVES"),
                    by_kernel={k: round(v.get("SQ_INSTS_VALU", 0.0)) for k, v in rec["by_kernel"].items()}, source=rec.get("source"),
                    note="VALU wave instructions of one receiver call (PMC) / RX-alone time of a call (measured here) against the VALU issue port of 1024 SIMDs: a "
                         "UTILISATION of that port, <= 1 by construction (round 5 also printed every instruction class against the same peak, which exceeded 1 because "
                         "SALU / LDS / VMEM / branch instructions issue from other ports in the same cycle: removed).  The decoder's other limit, the LDS pipe, and the "
                         "recursion's, dependent-chain latency, are in docs/KERNELS.md 7 and 9")
    except (OSError, ValueError, KeyError):
        return None


def promote_issue(roof, issue, dominant, why):
    """C3 / C5: the recursive QPSK chain is bound by instruction issue / the latency of its serial loops, not by HBM -- the issue
    figures become the headline of the sub-line's roofline (bound / achieved / peak / unit / frac), the HBM figures of the front-end kernel move
    into `hbm`, and `kernel` names the kernel that is dominant in the kernel trace (profiles/r05_kernel_trace_summary.md), not the HBM-facing one."""
    if not issue or issue.get("frac") is None:
        roof["dominant_kernel"] = dominant
        roof["dominant_kernel_note"] = why
        return roof
    hbm = {k: roof[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "kernel", "kernel_ms", "launches",
                                "algorithmic_bytes_per_launch", "algorithmic_bytes_per_sample", "whole_step", "flops", "note") if k in roof}
    out = {k: issue[k] for k in ("bound", "achieved", "peak", "unit", "frac")}
    out.update(kernel=dominant, kernel_note=why, issue=issue, hbm=hbm)
    return out


C4_CHANNEL = ("wideband noise (0.05 rms per component) + DMR-like 4FSK on channels 2, 17, 31, 45, 60 of every stream with SURVEY 8(d)'s timing: symbol clocks +-20 ppm, first symbol "
              "0.37 .. 0.81 symbol late (tests/sig.py make_4fsk clock_ppm / frac_delay); per stream a circular shift")


def c4_add_4fsk(iq, torch, seed):
    """DMR-like 4FSK carriers on five of the 64 channels of every wideband stream (the input of the C4 line: its parity_check then compares the dibits of real symbols
    under a sliding symbol phase, not only of noise); built once, untimed"""
    import sig
    B, n = iq.shape
    fs = 1.6e6
    t = torch.arange(n, device=iq.device, dtype=torch.float64)
    for k, c in enumerate((2, 17, 31, 45, 60)):
        x, _ = sig.make_4fsk(nsym=int(n / fs * 4800) - 8, seed=seed + k, amp=0.3, noise=0.0, fs=fs, clock_ppm=20.0 * (1 if k % 2 == 0 else -1), frac_delay=0.37 + 0.11 * k)
        base = np.zeros(n, np.complex64)
        base[:min(n, x.size)] = x[:n]
        f0 = c * 25000.0 if c <= 32 else (c - 64) * 25000.0
        ph = (f0 / fs) * t
        ph = (ph - torch.floor(ph)) * (2 * np.pi)
        carrier = torch.polar(torch.ones_like(ph, dtype=torch.float32), ph.to(torch.float32))
        xb = torch.from_numpy(base).to(iq.device)
        for b in range(B):
            iq[b] += torch.roll(xb, 7919 * b + 131 * k) * carrier
    return iq


def parity_check_c4(ch, iq, torch, nstreams=2, seed=6):
    """One call from a fresh state at the bench shape: every channel of `nstreams` random wideband streams against the oracle --
    int16 FM samples and 4FSK dibits bit for bit, rssi_tag_block values to 1e-4 dB (log10f) -- untimed."""
    import orc
    M = ch.cc
    ch.reset()
    out, cnt = ch.process(iq)
    rng = np.random.default_rng(seed)
    picks = sorted(int(b) for b in rng.choice(iq.shape[0], size=min(nstreams, iq.shape[0]), replace=False))
    for b in picks:
        ref, rref, dref = orc.demod_mmdvm_multi_full(iq[b].cpu().numpy(), M)
        o, c = out[b].cpu().numpy(), cnt[b].cpu().numpy()
        r, rc = ch.rssi[b].cpu().numpy(), ch.rssi_counts[b].cpu().numpy()
        d, dc = ch.dibits[b].cpu().numpy(), ch.fsk_counts[b].cpu().numpy()
        for k in range(M):
            ok = c[k] == ref.shape[1] and np.array_equal(o[k, :c[k]], ref[k])
            ok = ok and rc[k] == rref[k].size and np.allclose(r[k, :rc[k]], rref[k], rtol=0, atol=1e-4)
            ok = ok and dc[k, 2] == dref[k].size and np.array_equal(d[k, :dc[k, 2]], dref[k])
            if not ok:
                return dict(status="FAILED", stream=b, channel=k, streams=picks)
    return dict(status="bit-exact", streams=picks, compared="int16 FM samples and 4FSK dibits of all %d channels (bit for bit), RSSI tags (1e-4 dB), one call from a fresh state" % M, channel=C4_CHANNEL, against=PARITY_AGAINST,
                int16_per_channel=int(ref.shape[1]), dibit_bytes_per_channel=int(dref[0].size))


def run_c4(args, torch, q, ctx, dev, rank, world, steps=None, with_form2=True, check=False):
    """C4: multi-carrier MMDVM receiver, 64 x 25 kHz channels from 1.6 Msps wideband IQ (PFB channelizer + per-channel
    24/25 resampler, LPF, FM discriminator -> int16, RSSI tags and the 4FSK symbol tail).  Multi-GPU (PFB form): the CHANNELS
    are sharded -- every rank channelizes its own B / world wideband streams, one RCCL all_to_all_single per step moves each
    channel's 25 ksps samples to the rank that owns the channel, the per-channel chains run there ("strong" scaling: B is fixed).
    Also measured (single GPU): BASELINE configs[3] taken literally, form 2 = 64 frequency-translating FIRs 1:64 in front of the same
    per-channel chain -- the compute-bound way of producing the channels the PFB produces."""
    import copy
    args = copy.copy(args)
    if steps:
        args.steps = steps
    M, B, n = 64, args.batch or 64, (args.nsamp or (1 << 21)) // 64 * 64
    if M % world or B % world:
        raise SystemExit("c4: the number of ranks must divide 64 channels and the number of wideband streams")
    per = M // world
    g = torch.Generator(device=dev)
    g.manual_seed(7 + rank)
    from qradiolink_amd import sharding
    link_bytes = None
    if world == 1 and not args.cluster:
        iq = c4_add_4fsk(torch.view_as_complex(torch.randn((B, n, 2), generator=g, device=dev, dtype=torch.float32) * 0.05), torch, 40 + rank)
        ch = q.Channelizer(ctx, M, batch=B, max_chunk=n)
        ch.enable_4fsk()
        if args.legacy_pfb:
            ch.set_option(q.CHAN_OPT_LEGACY_PFB, args.legacy_pfb)
        step, sync, prof = (lambda: ch.process_async(iq)), ch.sync, ch
        handles = [ch]
    else:
        # SURVEY 8e, PFB form, driven through the C++ host class qrl_host::chan_cluster (qradiolink_amd/host/chan_cluster.*, libqrl_cluster.so):
        # every rank channelizes ITS B / world wideband streams (all 64 channels), ONE all-to-all per step (chan_exchange::all_to_all:
        # ncclAllToAll over xGMI on its own communicator) hands each rank the channels it owns of EVERY stream, the per-channel chains run
        # on the owner; ordering on the device (qrl_chan_stream_wait / qrl_chan_wait_for), no host synchronisation.  Per link and step:
        # (B / world) x (64 / world) x n / 64 cf32 items = 1 / world of a rank's input bytes.  (--cluster at N = 1: the same object
        # with a one-rank RCCL communicator -- what a one-GPU box can run of this path.)
        if getattr(args, "cluster_copy", False):
            os.environ["QRL_CLUSTER_COPY_AT_ONE_RANK"] = "1"
        Bl, n1 = B // world, n // M
        iq = c4_add_4fsk(torch.view_as_complex(torch.randn((Bl, n, 2), generator=g, device=dev, dtype=torch.float32) * 0.05), torch, 40 + rank)   # this rank's own inputs
        ex = sharding.Exchange.rccl(torch.distributed if world > 1 else None)
        cl = sharding.Cluster(ctx, ex, M, Bl, n)
        cl.tail.enable_4fsk()
        link_bytes = sharding.bytes_per_link_per_step(Bl, M, world, n1)
        ch = cl.front

        def step():
            cl.step_async(iq)

        def sync():
            cl.sync()
        prof, handles = cl.front, [cl.front, cl.tail]
    prof.profile(True)
    marks = StepMarks(torch, [x for hnd in handles for x in hnd.internal_streams], enabled=not args.no_marks)
    dt = timed_loop(step, sync, args, torch, dev, world, marks)
    per_kernel = None
    if world == 1 and not args.cluster:
        # every kernel of the call, HIP events on the stream it runs on (the warm-up calls are profiled too: same kernels, same shape)
        pk = prof.profile_read_kernels()
        _, _, kname = prof.profile_read()
        per_kernel = {(kname if k == "channelizer" else k): round(ms / max(cnt, 1), 4) for k, ms, cnt in pk if cnt}
        kms, launches = pk[0][1], pk[0][2]
    else:
        kms, launches, kname = prof.profile_read()
    launches_timed = args.steps
    kms = kms * launches_timed / max(launches, 1)          # (the warm-up calls were profiled too: same kernel, same shape)
    prof.profile(False)
    parity = parity_check_c4(ch, iq, torch) if (check and world == 1 and rank == 0 and not args.cluster) else None
    if world == 1 and not args.cluster:
        for h in handles:
            h.close()
    else:
        cl.close()
        ex.close()
    b_kernel = (B // world) if world > 1 else B             # wideband streams the channelizer of THIS rank processes per launch
    line = {"metric": "wideband IQ MSamples/sec through the C4 receiver", "value": round(B * n * args.steps / dt / 1e6, 1), "unit": "MS/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C4: 64 x 25 kHz MMDVM channels from 1.6 Msps IQ: PFB channelizer + FM int16 + RSSI + 4FSK tail",
                       "wideband_streams": B, "samples_per_stream_per_step": n, "channels_per_gpu": per,
                       "parallelism": ("channels sharded over ranks: every rank channelizes its %d wideband streams, one RCCL all_to_all_single per step moves "
                                       "the channel streams to their owners (%d bytes per link and step), per-channel chains on the owner" % (B // world, link_bytes))
                                      if (world > 1 or args.cluster) else "single GPU",
                       "bytes_per_link_per_step": link_bytes},
            "step_spread_ms": marks.spread()}
    ms_step = dt / args.steps * 1e3
    if link_bytes and world > 1:
        # how to read a SCALE point of this line: xGMI is point to point -- the world - 1 blocks of a rank leave on world - 1 links at once, so one
        # all-to-all takes ONE block's time; it hides under the 3-slot pipeline only while it is shorter than a rank's compute (= this line's single-GPU step / world
        # if the kernels scale).  profiles/r06_c4_emulated_ranks.jsonl has the same figures for 2 / 4 / 8 ranks from the one-device emulation.
        line["config"]["expected_exchange_ms"] = {"at_45_GBps_per_link": round(link_bytes / 45e9 * 1e3, 3), "at_64_GBps_per_link": round(link_bytes / 64e9 * 1e3, 3),
                                                  "note": "one block per link and step; compare with ms_per_step: a step cannot be shorter than the slower of compute and exchange"}
    if per_kernel:
        # the dominant kernel = the longest of the call (k_chan_tail since round 4); `frac` = the WHOLE chain's algorithmic bytes over the whole step
        # (VERDICT r4 #1a), the per-kernel figures beside it; C4 is bound by f32 instructions, not by HBM: the flop roofline says how far from THAT roof
        dom = max((k for k in per_kernel if not k.startswith("k_symsync")), key=lambda k: per_kernel[k])
        abytes = b_kernel * n * C4_BYTES
        ws = whole_step_obj(abytes, ms_step)
        traffic, src = pmc_traffic("c4", dom, not (args.batch or args.nsamp))
        chain_traffic, chain_by_kernel = pmc_chain_traffic("c4", not (args.batch or args.nsamp))
        flop = (C4_FLOP_PFB + C4_FLOP_TAIL) * b_kernel * n
        line["roofline"] = dict(
            bound="f32 instruction issue (VALU + matrix pipe); hbm figures for reference", achieved=ws["achieved"], peak=HBM_PEAK_GBPS, unit="GB/s", frac=ws["frac"],
            traffic=chain_traffic if chain_traffic else traffic, traffic_source=src, traffic_from=TRAFFIC_FROM % source_id() if (chain_traffic or traffic) else None,
            traffic_kernel=traffic, traffic_by_kernel=chain_by_kernel,
            traffic_note=("traffic = HBM bytes per step of the WHOLE chain (sum over its kernels, PMC); traffic_kernel = the dominant kernel's own. The chain hands two rings "
                          "from kernel to kernel -- 1.07 GB of 25 ksps channel samples (written by the channelizer, read with a 20 % halo by the per-channel kernel) and "
                          "0.5 GB of RRC output (written there, read by the symbol synchroniser) -- which is what separates it from the algorithmic bytes; at the step time "
                          "it is a quarter of the HBM roof, the chain is bound by f32 instruction issue (flops)") if chain_traffic else None,
            kernel=dom, kernel_ms=per_kernel[dom], launches=launches_timed,
            algorithmic_bytes_per_launch=abytes, algorithmic_bytes_per_sample=round(C4_BYTES, 3),
            kernels_ms=per_kernel,
            kernel_fracs={k: round(abytes / (v * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) for k, v in per_kernel.items() if v > 0},
            whole_step=ws,
            flops=dict(bound="f32", achieved=round(flop / (ms_step * 1e-3) / 1e12, 2), peak=F32_PEAK_TFLOPS, unit="TFLOP/s",
                       frac=round(flop / (ms_step * 1e-3) / 1e12 / F32_PEAK_TFLOPS, 4),
                       algorithmic_flop_per_wideband_sample=round(C4_FLOP_PFB + C4_FLOP_TAIL, 1),
                       channelizer_flop_per_sample=C4_FLOP_PFB, per_channel_chain_flop_per_sample=round(C4_FLOP_TAIL, 1),
                       by_kernel={k: round((C4_FLOP_PFB if "pfb" in k else C4_FLOP_TAIL) * b_kernel * n / (v * 1e-3) / 1e12, 2)
                                  for k, v in per_kernel.items() if v > 0 and not k.startswith("k_symsync")},
                       note="flop of the arithmetic contract (oracle) per wideband sample x samples / step time / 157.3 TFLOP/s; by_kernel: that kernel's flop / its own duration"),
            note="frac = whole chain over the whole step (the kernels of consecutive calls overlap on three streams); kernel = the longest kernel of a call, "
                 "kernel_fracs = the chain's algorithmic bytes over each kernel's own duration (HIP events on its stream, beside whatever shares the chip)")
    else:
        line["roofline"] = roofline_obj(kname, kms, launches_timed, b_kernel * n * C4_BYTES, round(C4_BYTES, 3),
                                        "the channelizer reads the wideband input once and writes the 64 channel rings; whole chain: %.1f GB/s of algorithmic bytes"
                                        % (B * n * C4_BYTES * args.steps / dt / 1e9), name="c4", default_shape=not (args.batch or args.nsamp), ms_per_step=ms_step)
    if parity:
        line["parity_check"] = parity
    if world == 1 and with_form2 and not args.cluster:
        # BASELINE configs[3] literally: 64 freq-xlating FIRs (2181 taps, 1:64) -- compute bound (34 MAC per input sample and channel)
        B2, n2 = max(1, B // 8), n // 4
        ch2 = q.Channelizer(ctx, M, batch=B2, max_chunk=n2, form=2)
        ch2.enable_4fsk()
        iq2 = iq[:B2, :n2].contiguous()
        a2 = copy.copy(args)
        a2.steps, a2.warmup = max(2, args.steps // 5), 1
        ch2.profile(True)
        dt2 = timed_loop(lambda: ch2.process_async(iq2), ch2.sync, a2, torch, dev, world)
        kms2, l2, kname2 = ch2.profile_read()
        ch2.close()
        flop = 2.0 * 2 * 2181 / 64 * 64            # real x complex MAC = 4 flop; taps / decimation MACs per channel, 64 channels
        line["freq_xlating_form"] = {
            "workload": "configs[3] literal: 64 x (rotator + rational_resampler_ccf(1, 64, 2181 taps)) + the same per-channel chain + 4FSK tail (form 2)",
            "value": round(B2 * n2 * a2.steps / dt2 / 1e6, 1), "unit": "MS/s", "steps": a2.steps, "ms_per_step": round(dt2 / a2.steps * 1e3, 3),
            "wideband_streams": B2, "samples_per_stream_per_step": n2, "kernel": kname2, "kernel_ms_per_step": round(kms2 / max(l2, 1), 3),
            "note": "compute bound, not an HBM kernel (no hbm roofline reported): %.0f flop per wideband sample in the 64 decimators = %.1f TFLOP/s on the "
                    "f32 matrix pipe (peak 157); 64 launches that each re-read the input -- the PFB form above is the production path"
                    % (flop, flop * B2 * n2 * l2 / (kms2 * 1e-3) / 1e12 if kms2 > 0 else 0.0)}
    return line


C5_RX_BYTES = 8.0 + (500000 * 8 + 250000 * 8 + 250000) / 1e6   # C3's ports at 1 Msps: 14.25 B per RX sample; TX: 8 B written per sample


def parity_check_c5(dem, mod, iq, data, tx_out, torch, nstreams=3, seed=8):
    """C5 at the bench shape, untimed: one RX call from a fresh state -- bits and port 0 of `nstreams` random streams against the
    oracle's gr_demod_qpsk chain -- and one TX call from a fresh state -- the 1 Msps samples of `nstreams` streams against the
    oracle's gr_mod_qpsk chain, float for float."""
    import orc
    dem.reset()
    mod.reset()
    out = dem.process(iq)
    mod.process_async(data, out=tx_out)
    mod.sync()
    cnt = out["counts"].cpu().numpy()
    rng = np.random.default_rng(seed)
    picks = sorted(int(b) for b in rng.choice(iq.shape[0], size=min(nstreams, iq.shape[0]), replace=False))
    for b in picks:
        ref = oracle_demod("qpsk250k", iq[b].cpu().numpy(), 1000000, 0.0)
        ok = np.array_equal(out["bits_a"][b, :cnt[b, 2]].cpu().numpy(), ref["bits_a"])
        got_f = out["filtered"][b, :cnt[b, 0]].cpu().numpy().view(np.float32) + np.float32(0)
        want_f = ref["filtered"].view(np.float32) + np.float32(0)
        ok = ok and got_f.size == want_f.size and np.array_equal(got_f.view(np.uint32), want_f.view(np.uint32))
        tx_ref = orc.mod_qpsk(data[b].cpu().numpy(), sps=4, filter_width=160000)
        tx_got = tx_out[b].cpu().numpy()
        g, w = tx_got.view(np.float32) + np.float32(0), tx_ref.view(np.float32) + np.float32(0)
        ok = ok and g.size == w.size and np.array_equal(g.view(np.uint32), w.view(np.uint32))
        if not ok:
            return dict(status="FAILED", stream=b, streams=picks)
    return dict(status="bit-exact", streams=picks, compared="RX: bits and port 0 (filtered); TX: every 1 Msps sample; one call each from a fresh state",
                channel="SURVEY 8(d) (tests/sig.py SPEC): fractional delay 0.37 sample, clock error +20 ppm, CFO, AWGN at Es/N0 12 dB; every stream the same waveform", against=PARITY_AGAINST,
                bits_per_stream=int(cnt[picks[0], 2]), tx_samples_per_stream=int(tx_out.shape[1]))


def run_c5(args, torch, q, ctx, dev, rank, world, steps=None, check=False):
    """C5: full duplex -- QPSK-250k modulator and QPSK-250k demodulator handles on their own HIP streams, calls interleaved without
    synchronisation (BASELINE config 5; reference src/radiocontroller.cpp:2043-2078 runs the two top blocks concurrently)."""
    import sig
    # 16 384 streams x 16 384 samples per call: the recursive QPSK chain is a serial walk over a call's samples of one stream (about
    # 0.2 us per 500 ksps sample whatever the batch), so the same number of samples as more, shorter streams is what fills the chip
    B = args.batch or 16384
    n = (args.nsamp or (1 << 14)) & ~1
    nbytes = n // 32                                  # the TX produces as many 1 Msps samples as the RX consumes
    base, _ = sig.make_stream("qpsk250k", nframes=3, device_rate=1000000, seed=3 + rank, amp=0.05, impair=sig.SPEC)   # SURVEY 8(d)'s channel
    base = np.tile(base, -(-n // base.size))[:n]
    iq = torch.from_numpy(base).to(dev).repeat(B, 1).contiguous()
    g = torch.Generator(device=dev)
    g.manual_seed(11 + rank)
    data = torch.randint(0, 256, (B, nbytes), generator=g, device=dev, dtype=torch.uint8)
    import copy
    args = copy.copy(args)
    if steps:
        args.steps = steps
    dem = q.Demod(ctx, 26, batch=B, max_chunk=n)
    if getattr(args, "no_grouped", False):
        dem.set_option(q.OPT_GROUPED, 0)
    tx_stream = torch.cuda.Stream()                  # the modulator handle runs on a stream this script can order
    mod = q.Mod(ctx, 26, batch=B, max_bytes=nbytes, stream=tx_stream.cuda_stream)
    tx_out = torch.empty((B, nbytes * mod.spb), dtype=torch.complex64, device=dev)
    main_stream = torch.cuda.ExternalStream(dem.stream)
    lockstep = not getattr(args, "free_tx", False)

    def both():
        # one TX chunk per RX chunk, as two flowgraphs clocked by the same 1 Msps hardware are: TX call k is ordered (on the device, an
        # event; the host never waits) behind the receiver's front end of call k, so it shares the chip with the recursion / decoder
        # phase.  Unordered (--free-tx) the modulator stream runs ~ 60 calls ahead of the receiver -- samples a radio could not have
        # sent yet -- and its tens of thousands of small workgroups keep the recursion kernel's 137 KB workgroups waiting for a CU.
        dem.process_async(iq)
        if lockstep:
            e = torch.cuda.Event()
            e.record(main_stream)
            tx_stream.wait_event(e)
        mod.process_async(data, out=tx_out)

    def sync():
        mod.sync()
        dem.sync()
    # step marks on the receiver's streams only (events order nothing; the decoder of a call is launched one call late in the grouped
    # order, so its stream's mark belongs to the call before -- in steady state the intervals are the step period all the same)
    marks = StepMarks(torch, dem.internal_streams, enabled=not args.no_marks)
    dt = timed_loop(both, sync, args, torch, dev, world, marks)
    dem.profile(True)
    dt_rx = timed_loop(lambda: dem.process_async(iq), sync, args, torch, dev, world)
    kms, launches, kname = dem.profile_read()
    dem.profile(False)
    dt_tx = timed_loop(lambda: mod.process_async(data, out=tx_out), sync, args, torch, dev, world)
    parity = parity_check_c5(dem, mod, iq, data, tx_out, torch) if (check and rank == 0) else None
    dem.close()
    mod.close()
    tot = float(B) * n * args.steps * world
    line = {"metric": "IQ MSamples/sec through RX demod chain (with the TX chain running concurrently)", "value": round(tot / dt / 1e6, 1),
            "unit": "MS/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C5: full duplex QPSK-250k TX + RX at 1 Msps on two HIP streams", "streams_per_gpu": B,
                       "samples_per_stream_per_step": n, "tx_msps_concurrent": round(tot / dt / 1e6, 1),
                       "rx_alone_ms_per_step": round(dt_rx / args.steps * 1e3, 3), "tx_alone_ms_per_step": round(dt_tx / args.steps * 1e3, 3)},
            "roofline": roofline_obj(kname, kms, launches, B * n * C5_RX_BYTES, C5_RX_BYTES,
                                     "RX front end (1:2 decimator + RRC) timed in the RX-alone pass; duplex chain: %.1f GB/s of algorithmic bytes (RX %.2f + TX 8 B per sample); TX alone writes %.1f GB/s"
                                     % (tot * (C5_RX_BYTES + 8.0) / dt / 1e9, C5_RX_BYTES, tot * 8.0 / dt_tx / 1e9),
                                     name="c5", default_shape=not (args.batch or args.nsamp)),
            "step_spread_ms": marks.spread()}
    issue = issue_roofline("c5", dt_rx / args.steps, default_shape=not (args.batch or args.nsamp))
    line["roofline"]["whole_step"] = whole_step_obj(tot / args.steps * (C5_RX_BYTES + 8.0), dt / args.steps * 1e3)
    line["roofline"] = promote_issue(line["roofline"], issue, "k_fec, k_qpsk_pipe4",
                                     "a receiver call is k_dec2_fir -> k_qpsk_pipe4 (the serial QPSK recursion) || k_fec (Viterbi decoder of the call before): the two "
                                     "longest kernels of the trace -- the recursion bound by the latency of its dependent chain, the decoder by the VALU port; the HBM-facing front end k_dec2_fir is in `hbm`")
    if parity:
        line["parity_check"] = parity
    return line


def cpu_baseline(name, cores, budget_s=8.0):
    """The oracle's CPU restatement of the reference flowgraph on a bounded sample of the same workload.  Three figures:
    value          all host cores, OpenMP over independent streams, decimators as an AVX2 dot product (VOLK-like; what a
                   GNU Radio + VOLK build would vectorise); one stream per core like one flowgraph per core
    single_thread  the same on ONE core (= one reference flowgraph, ignoring GNU Radio's block-per-thread overlap)
    scalar_port    the bit-exact checker itself (scalar fmaf chains in the GPU's summation order) on all cores"""
    import orc
    import sig
    label, mode, modem, rate, offset, _, _, omode, _ = WORKLOADS[name]
    per = (1 << 22) if rate >= 2000000 else (1 << 20)
    base, _ = sig.make_stream(mode, nframes=2, device_rate=rate, rx_offset_hz=offset, seed=99, amp=0.05)
    base = np.tile(base, -(-per // base.size))[:per]

    def timed(nstreams, threads, impl, budget):
        iq = np.stack([np.roll(base, 977 * b) for b in range(nstreams)]).astype(np.complex64)
        orc.lib.orc_set_decim_impl(impl)
        total, reps = 0.0, 0
        while total < budget and reps < 5000:
            secs, _ = orc.batch_rx(omode, iq, rate, offset, threads)
            total += secs
            reps += 1
        orc.lib.orc_set_decim_impl(0)
        return reps * nstreams * per / total / 1e6, reps, total
    v_all, reps, tot = timed(cores, cores, 1, budget_s)
    v_one, _, _ = timed(1, 1, 1, budget_s / 4)
    v_port, _, _ = timed(cores, cores, 0, budget_s / 2)
    # SURVEY 8(d) variant (ii), GNU Radio's thread-per-block scheduler: every block of the flowgraph on its own thread, streams
    # flowing through rings.  In steady state such a pipeline moves one stream at the pace of its SLOWEST block, so the figure is
    # derived from the measured single-thread run time of every block of one chain run (oracle/orc_trace.c block timing), best of 3:
    # samples / max(block seconds).  It needs as many cores as the flowgraph has blocks and is an upper bound (no ring overhead).
    orc.lib.orc_set_decim_impl(1)
    runs = [orc.block_times(omode, base, rate, offset) for _ in range(3)]
    orc.lib.orc_set_decim_impl(0)
    bt = min(runs, key=lambda r: max(t for _, t in r))
    slow = max(bt, key=lambda v: v[1])
    tpb = dict(value=round(per / slow[1] / 1e6, 3), unit="MS/s", blocks=len(bt), slowest_block=slow[0],
               slowest_block_share=round(slow[1] / sum(t for _, t in bt), 3),
               note="one stream, thread-per-block pipeline bound = samples / run time of the slowest block (measured per block, "
                    "single thread each); x streams when cores >= blocks x streams")
    # ... and EMULATED (round 6; C1's chain at 1 Msps only): one thread per block of the 2FSK receiver, bounded queues in between, whole
    # streams as the items that flow (oracle/orc_pipeline.c).  Its checksum has to equal the plain chain's.
    tpe = None
    if name == "c1":
        ns = max(12, min(64, int(budget_s / 4 * tpb["value"] * 1e6 / per)))
        iq = np.stack([np.roll(base, 977 * b) for b in range(ns)]).astype(np.complex64)
        orc.lib.orc_set_decim_impl(1)
        secs, reps_p, share, ok = 0.0, 0, 0.0, True
        _, chk_ref = orc.batch_rx(omode, iq, rate, offset, cores)
        while secs < budget_s / 4 and reps_p < 200:
            t, chk, busy = orc.pipeline_rx_2fsk1k(iq, offset)
            secs += t; reps_p += 1; share = max(share, max(busy) / t); ok = ok and chk == chk_ref
        orc.lib.orc_set_decim_impl(0)
        tpe = dict(value=round(reps_p * ns * per / secs / 1e6, 3), unit="MS/s", threads=len(busy) + 2, streams=ns, passes=reps_p, checksum_equals_chain=bool(ok),
                   busiest_stage_share=round(share, 3),
                   note="%d passes of %d streams x %d samples through 11 block threads + source + sink (%.1f s): one flowgraph's rate; a host runs cores / threads of them side by side" % (reps_p, ns, per, secs))
    return dict(value=round(v_all, 3), unit="MS/s", cores=cores, kind="port",
                single_thread=round(v_one, 3), scalar_port=round(v_port, 3), thread_per_block_model=tpb, thread_per_block_emulated=tpe,
                sample="%d passes over %d streams x %d samples of the %s workload (%.1f s of CPU wall time); oracle/liborc.so "
                       "(C, -O3 -mavx2 -mfma, OpenMP over streams) with the decimating FIRs as AVX2 dot products "
                       "(orc_decim_fir_ccf_simd); GNU Radio / VOLK itself is not installable here"
                       % (reps, cores, per, name.upper(), tot))


def _cpu_threads(fn, items, threads):
    """run fn(item) for every item on `threads` host threads (the oracle's C functions release the GIL under ctypes)"""
    from concurrent.futures import ThreadPoolExecutor
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(fn, items))
    return time.perf_counter() - t0


def cpu_baseline_c4(cores, budget_s=4.0):
    """C4 on the host: the oracle's whole multi-carrier receiver (PFB + 64 per-channel chains + 4FSK tails) on `cores` threads, one
    wideband stream per thread (the reference runs ONE such flowgraph; GNU Radio would spread its blocks over threads)."""
    import orc
    n = 1 << 18
    rng = np.random.default_rng(3)
    xs = [(0.05 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64) for _ in range(cores)]
    one = _cpu_threads(lambda x: orc.demod_mmdvm_multi_full(x, 64), xs[:1], 1)
    total, reps = 0.0, 0
    while total < budget_s and reps < 1000:
        total += _cpu_threads(lambda x: orc.demod_mmdvm_multi_full(x, 64), xs, cores)
        reps += 1
    return dict(value=round(reps * cores * n / total / 1e6, 3), unit="MS/s", cores=cores, kind="port", single_thread=round(n / one / 1e6, 3),
                sample="%d passes over %d wideband streams x %d samples of the C4 workload (%.1f s of CPU wall time); oracle/liborc.so scalar restatement "
                       "(direct-sum DFT as the contract defines it; upstream runs FFTW there), one stream per thread" % (reps, cores, n, total))


def cpu_baseline_c5(cores, budget_s=4.0):
    """C5 on the host: the oracle's QPSK-250k receiver on `cores` threads (one stream each) WHILE the same number of streams is modulated
    on the same threads afterwards -- value = RX samples / (RX time + TX time of as many samples), i.e. both directions on the same cores."""
    import orc
    import sig
    n = 1 << 18
    base, _ = sig.make_stream("qpsk250k", nframes=3, device_rate=1000000, seed=98, amp=0.05)
    base = np.tile(base, -(-n // base.size))[:n]
    rng = np.random.default_rng(4)
    xs = [np.roll(base, 911 * b).astype(np.complex64) for b in range(cores)]
    ds = [rng.integers(0, 256, n // 32, dtype=np.uint8) for _ in range(cores)]
    rx = lambda x: orc.demod_qpsk(orc.frontend(x, 1000000, 0.0), sps=2, filter_width=160000)
    total_rx = total_tx = 0.0
    reps = 0
    while total_rx + total_tx < budget_s and reps < 1000:
        total_rx += _cpu_threads(rx, xs, cores)
        total_tx += _cpu_threads(orc.mod_qpsk, ds, cores)
        reps += 1
    return dict(value=round(reps * cores * n / (total_rx + total_tx) / 1e6, 3), unit="MS/s", cores=cores, kind="port",
                rx_only=round(reps * cores * n / total_rx / 1e6, 3), tx_only=round(reps * cores * n / total_tx / 1e6, 3),
                sample="%d passes over %d streams x %d samples each way of the C5 workload (%.1f s of CPU wall time); oracle/liborc.so scalar restatement, "
                       "one stream per thread" % (reps, cores, n, total_rx + total_tx))


def respawn_under_torchrun(args):
    """python bench.py --gpus N started without a launcher: become N ranks (one per GPU) on this node."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c1", choices=sorted(WORKLOADS) + ["c4", "c5"])
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--nsamp", type=int, default=0)
    ap.add_argument("--pad", type=int, default=0, help="extra samples of row pitch of the input batch (even)")
    ap.add_argument("--no-extra", action="store_true", help="only the timed workload: no stand-alone pass, parity check, C2 line, CPU baseline")
    ap.add_argument("--overlap", action="store_true", help="(the library default since round 3; kept for the tools/ scripts)")
    ap.add_argument("--no-overlap", action="store_true", help="C1 only: QRL_OPT_OVERLAP = 0 (the kernels of a call strictly one after the other)")
    ap.add_argument("--free-tx", action="store_true", help="C5 only: the modulator stream not ordered behind the receiver's front end (it then runs tens of calls ahead)")
    ap.add_argument("--no-grouped", action="store_true", help="C5 only: QRL_OPT_GROUPED = 0 (three free-running streams instead of front end -> recursion || decoder)")
    ap.add_argument("--fll-slim", action="store_true", help="tuning, c1: QRL_OPT_FLL_SLIM = 1 (single-wave FLL workgroups)")
    ap.add_argument("--cluster", action="store_true", help="c4: drive the channel-sharded path (qrl_host::chan_cluster + RCCL all-to-all) also at N = 1")
    ap.add_argument("--cluster-copy", action="store_true", help="c4 --cluster at N = 1: keep the one-rank ncclAllToAll (a 1 GB device copy per step that moves nothing; by default one rank reads the channelizer's output in place)")
    ap.add_argument("--no-marks", action="store_true", help="no per-step completion events (step_spread_ms = null)")
    ap.add_argument("--check", action="store_true", help="with --no-extra: still run the parity check against the oracle at the bench shape")
    ap.add_argument("--legacy-pfb", type=int, default=0, help="c4: QRL_CHAN_OPT_LEGACY_PFB = 1 (the general-M channelizer kernel at the 64-channel shape)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_torchrun(args)

    import torch
    import qradiolink_amd as q

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but the launcher started %d ranks; reporting n_gpus = %d" % (args.gpus, world, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl", device_id=dev)
    ctx = q.Context(local)

    def finish(line):
        if rank == 0 and line is not None:
            # (RCCL writes its version banner through C stdio, which is fully buffered when stdout is a file: flush it first so that
            #  the JSON line is the LAST line of the output)
            try:
                import ctypes
                ctypes.CDLL(None).fflush(None)
            except OSError:
                pass
            sys.stdout.flush()
            print(json.dumps(line), flush=True)
        ctx.close()
        if world > 1:
            torch.distributed.barrier()   # rank 0 may be behind by the CPU baseline: leave together
            torch.distributed.destroy_process_group()

    extra_ok = not args.no_extra
    ncores = min(os.cpu_count() or 1, 16)
    if args.config in ("c4", "c5"):
        kw = dict(with_form2=extra_ok) if args.config == "c4" else {}
        line = (run_c4 if args.config == "c4" else run_c5)(args, torch, q, ctx, dev, rank, world, check=extra_ok or args.check, **kw)
        if extra_ok and rank == 0 and world == 1:
            line["cpu_baseline"] = (cpu_baseline_c4 if args.config == "c4" else cpu_baseline_c5)(ncores)
        finish(line)
        if rank == 0 and line.get("parity_check", {}).get("status", "bit-exact") != "bit-exact":
            raise SystemExit("bench.py: parity check against the oracle FAILED at the bench shape: %r" % (line["parity_check"],))
        return
    main_r = run_workload(args.config, args, torch, q, ctx, dev, rank, world, overlap=False if args.no_overlap else None, check=extra_ok or args.check)
    # C1 runs in the library's default mode (the FLL / discriminator kernels of call k share the GPU with the front end of call
    # k + 1: more whole-chain throughput, but the front-end kernel stretches).  The serial order -- where the front-end kernel has the
    # chip to itself -- is measured in a second short pass for the record.
    ovl = run_workload("c1", args, torch, q, ctx, dev, rank, world, overlap=False, steps=min(args.steps, 20)) \
        if (extra_ok and args.config == "c1" and not args.no_overlap) else None
    extra = run_workload("c2", args, torch, q, ctx, dev, rank, world, steps=min(args.steps, 50), check=True) if (extra_ok and args.config == "c1") else None
    extra3 = run_workload("c3", args, torch, q, ctx, dev, rank, world, steps=min(args.steps, 30), check=True) if (extra_ok and args.config == "c1") else None
    extra4 = run_c4(args, torch, q, ctx, dev, rank, world, steps=min(args.steps, 20), check=True) if (extra_ok and args.config == "c1" and world == 1) else None
    extra5 = run_c5(args, torch, q, ctx, dev, rank, world, steps=min(args.steps, 20), check=True) if (extra_ok and args.config == "c1" and world == 1) else None
    # (the CPU baseline is a property of the box, not of the job: rank 0 at N = 1 only, as the contract says)
    all_lines = extra_ok and rank == 0 and world == 1 and args.config == "c1"
    base = cpu_baseline(args.config, ncores) if (extra_ok and rank == 0 and world == 1) else None
    base2 = cpu_baseline("c2", ncores, budget_s=3.0) if all_lines else None
    base3 = cpu_baseline("c3", ncores, budget_s=3.0) if all_lines else None
    base4 = cpu_baseline_c4(ncores, budget_s=3.0) if all_lines else None
    base5 = cpu_baseline_c5(ncores, budget_s=3.0) if all_lines else None

    line = None
    if rank == 0:
        def roof(r):
            # HBM traffic per launch of the dominant kernel: pmc_traffic() above
            traffic, src = pmc_traffic(r["name"], r["kernel"], r["default_shape"])
            d = dict(bound="hbm", achieved=round(r["achieved_gbps"], 1), peak=HBM_PEAK_GBPS, unit="GB/s",
                     frac=round(r["achieved_gbps"] / HBM_PEAK_GBPS, 4), traffic=traffic, traffic_source=src,
                     traffic_from=TRAFFIC_FROM % source_id() if traffic else None, kernel=r["kernel"],
                     kernel_ms=round(r["kernel_ms"], 4), launches=r["launches"],
                     algorithmic_bytes_per_launch=r["bytes_per_launch"], algorithmic_bytes_per_sample=r["bytes_per_sample"],
                     whole_step=whole_step_obj(r["bytes_per_launch"], r["ms_per_step"]))
            # the compute roof of the same kernel: contract flop of the front end per input sample (real tap x complex sample = 4 flop per tap and output -> 4 nt / D
            # per input sample, + the rotator's complex product) over the kernel's duration against the f32 peak (matrix = packed vector rate: the two share the FMA
            # lanes, profiles/r05_sq_front_end_counters.txt: matrix pipe + other VALU = 98 % of C2's kernel cycles, 85 % of C3's -- busy cycles that add, padding included)
            fl = FRONT_END_FLOP.get(r["name"])
            if fl and r["kernel_ms"] > 0:
                ach_f = fl * r["batch"] * r["nsamp"] / (r["kernel_ms"] * 1e-3) / 1e12
                d["flops"] = dict(bound="f32", achieved=round(ach_f, 2), peak=F32_PEAK_TFLOPS, unit="TFLOP/s", frac=round(ach_f / F32_PEAK_TFLOPS, 4),
                                  algorithmic_flop_per_sample=round(fl, 1),
                                  note="contract flop (4 x taps / decimation + 6 for the rotator, per input sample) / kernel time; the kernel also executes the zero "
                                       "padding of its tiles (C2 / C3: 42 lags as 48, 25 phases as 28) and the exact-NCO table arithmetic, which are not counted here")
            if ovl and r["name"] == "c1":
                d["serial_mode"] = dict(kernel_ms=round(ovl["kernel_ms"], 4), achieved=round(ovl["achieved_gbps"], 1),
                                        frac=round(ovl["achieved_gbps"] / HBM_PEAK_GBPS, 4), ms_per_step=round(ovl["ms_per_step"], 3),
                                        value=round(ovl["msps"], 1),
                                        note="QRL_OPT_OVERLAP = 0: same workload, the kernels of a call one after the other -- the front-end kernel alone on the chip")
                d["note"] = ("default mode = QRL_OPT_OVERLAP 1: the FLL / discriminator / symbol-sync / decoder kernels of call k run beside this "
                             "kernel of call k + 1, so its launches are longer than in serial_mode and the step is shorter")
            if r["name"] == "c3":
                issue = issue_roofline("c3", r["ms_per_step"] * 1e-3, r["default_shape"])
                d = promote_issue(d, issue, "k_qpsk_pipe4",
                                  "the step IS this kernel's serial latency (8192 samples per stream and call through the first Costas loop's dependent chain, "
                                  "~ 474 cycles per sample whatever the batch); the HBM-facing front end k_decim_pm runs beside it and is in `hbm`")
            return d
        line = {
            "source_id": source_id(),
            "metric": "IQ MSamples/sec through RX demod chain", "value": round(main_r["msps"], 1), "unit": "MS/s",
            "n_gpus": world, "steps": main_r["steps"], "warmup": args.warmup, "ms_per_step": round(main_r["ms_per_step"], 3),
            "step_spread_ms": main_r["spread"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": main_r["label"], "streams_per_gpu": main_r["batch"],
                       "samples_per_stream_per_step": main_r["nsamp"], "device_samp_rate": main_r["rate"],
                       "parallelism": "streams sharded over ranks, no collective",
                       "decoded_bits_per_stream_last_step": main_r["bits_per_stream"]},
            "roofline": roof(main_r),
        }
        if main_r["parity"]:
            line["parity_check"] = main_r["parity"]
        if base:
            line["cpu_baseline"] = base
        for key, ex, cb in (("c2", extra, base2), ("c3", extra3, base3)):
            if ex:
                line[key] = {"workload": ex["label"], "value": round(ex["msps"], 1), "unit": "MS/s", "steps": ex["steps"],
                             "ms_per_step": round(ex["ms_per_step"], 3), "step_spread_ms": ex["spread"], "streams_per_gpu": ex["batch"],
                             "samples_per_stream_per_step": ex["nsamp"], "roofline": roof(ex)}
                if ex["parity"]:
                    line[key]["parity_check"] = ex["parity"]
                if cb:
                    line[key]["cpu_baseline"] = cb
        for key, ex, cb in (("c4", extra4, base4), ("c5", extra5, base5)):
            if ex:
                line[key] = {k: ex[k] for k in ("value", "unit", "steps", "ms_per_step", "step_spread_ms", "config", "roofline", "parity_check", "freq_xlating_form") if k in ex}
                if cb:
                    line[key]["cpu_baseline"] = cb
    finish(line)
    if rank == 0:
        failed = [(k, v["parity_check"]) for k, v in [("c1", line)] + [(k, line[k]) for k in ("c2", "c3", "c4", "c5") if k in line]
                  if v.get("parity_check") and v["parity_check"]["status"] != "bit-exact"]
        if failed:
            raise SystemExit("bench.py: parity check against the oracle FAILED at the bench shape: %r" % (failed,))


if __name__ == "__main__":
    main()
