#!/usr/bin/env python3
"""VERDICT r5 #7 -- the multi-GPU step of C4 at the N-rank SHAPE on ONE device (no 8-GPU node in this pool).

N chan_cluster objects of one process behind qrl_host::local_group (a real all-to-all among their buffers: device copies with an all-to-all's
dependency structure), each with 64 / N wideband streams and 64 / N channels -- together the whole C4 job (64 streams x 64 channels) on one GPU.
Measured, with the steps queued back to back and no host synchronisation:
  plain      the single-handle receiver (qrl_chan_process), the figure of the c4 bench line
  emulated   N emulated ranks, exchange included
  no_copies  the same with the exchange's copies skipped (dependencies kept): what the data movement costs the step
  exchange   the exchange alone (N^2 block copies per step), back to back
`hidden` = 1 - (emulated - no_copies) / exchange: the share of the exchange that disappears under the 3-slot pipeline.
Then what a real N-GPU node moves: bytes per link and step, and the time at an assumed xGMI all-to-all rate.

usage: python tools/c4_emulated_ranks.py [--ranks 8] [--steps 30] [--n 2097152] [--out file.json]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(torch, fn, sync, steps, warmup=3):
    for _ in range(warmup):
        fn()
    sync()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    import time
    h0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    return (time.perf_counter() - h0) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--n", type=int, default=1 << 21)
    ap.add_argument("--streams", type=int, default=64)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import torch
    import qradiolink_amd as q
    from qradiolink_amd import sharding
    M, B, W, n = 64, a.streams, a.ranks, a.n
    assert B % W == 0 and M % W == 0
    dev = torch.device("cuda:0")
    ctx = q.Context(0)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    iq = torch.view_as_complex(torch.randn((B, n, 2), generator=g, device=dev, dtype=torch.float32) * 0.05)
    res = {"shape": {"ranks": W, "wideband_streams": B, "streams_per_rank": B // W, "channels_per_rank": M // W, "samples_per_stream_per_step": n}}
    # plain handle
    ch = q.Channelizer(ctx, M, batch=B, max_chunk=n)
    ch.enable_4fsk()
    res["plain_ms"] = round(timed(torch, lambda: ch.process_async(iq), ch.sync, a.steps), 3)
    ch.close()
    # emulated ranks
    em = sharding.EmulatedRanks(ctx, W, M, B // W, n)
    for c in em.cls:
        c.tail.enable_4fsk()
    res["emulated_ms"] = round(timed(torch, lambda: em.step(iq), em.sync, a.steps), 3)
    b0 = em.group.bytes_moved()
    em.step(iq); em.sync()
    res["exchange_bytes_per_step_all_ranks"] = em.group.bytes_moved() - b0
    em.group.skip_copies(True)
    res["no_copies_ms"] = round(timed(torch, lambda: em.step(iq), em.sync, a.steps), 3)
    em.group.skip_copies(False)
    # the exchange alone: the group's copies between the ranks' buffers, back to back on their exchange streams
    send = [torch.empty((W, (B // W) * (M // W) * (n // M), 2), dtype=torch.float32, device=dev) for _ in range(W)]
    recv = [torch.empty_like(s) for s in send]
    grp = sharding.LocalGroup(W)
    mem = [grp.member(r) for r in range(W)]
    streams = [torch.cuda.Stream() for _ in range(W)]

    def xchg():
        for r in range(W):
            mem[r].all_to_all(send[r], recv[r], stream=streams[r].cuda_stream)

    def xsync():
        for s in streams:
            s.synchronize()
    res["exchange_alone_ms"] = round(timed(torch, xchg, xsync, a.steps), 3)
    for m_ in mem:
        m_.close()
    grp.close()
    em.close()
    ctx.close()
    cost = res["emulated_ms"] - res["no_copies_ms"]
    res["exchange_cost_in_step_ms"] = round(cost, 3)
    res["hidden"] = round(1.0 - cost / res["exchange_alone_ms"], 3) if res["exchange_alone_ms"] > 0 else None
    res["emulation_overhead_vs_plain"] = round(res["no_copies_ms"] / res["plain_ms"] - 1.0, 3)
    # what a real node moves: every rank sends (W - 1) blocks of streams_per_rank x channels_per_rank x n / M cf32 items, one per xGMI link
    per_link = sharding.bytes_per_link_per_step(B // W, M, W, n // M)
    res["real_node"] = {
        "bytes_per_link_per_step": per_link,
        "bytes_sent_per_rank_per_step": per_link * (W - 1),
        "compute_per_rank_ms_if_it_scales": round(res["plain_ms"] / W, 3),
        "xgmi_ms_at_45_GBps_per_link": round(per_link / 45e9 * 1e3, 3),
        "xgmi_ms_at_64_GBps_per_link": round(per_link / 64e9 * 1e3, 3),
        "note": "xGMI is point to point: the W - 1 blocks of a rank leave on W - 1 different links at once, so the all-to-all takes one block's time "
                "(MI355X_MICROARCH.md: 7 links x ~153 GB/s bidirectional; 45-64 GB/s per direction is what an RCCL all-to-all reaches); it hides under the "
                "3-slot pipeline when it is shorter than a rank's compute",
    }
    line = json.dumps(res)
    print(line)
    if a.out:
        with open(a.out, "w") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()
