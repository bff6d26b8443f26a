"""N>1 path on CPU: world_size-2 gloo run of the stream sharding + result gather used by bench.py
and the multi-channel path (shard by stream/channel, no data-path collective).
The per-stream work is done here by the CPU oracle as a stand-in for the device pipeline (there is no GPU in the CPU test run);
tests/test_gpu_sharding.py runs the same sharding with the HIP pipeline: two handles for stream shards, two processes + a
broadcast of the wideband input for channel shards."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import orc
import sig
from qradiolink_amd import sharding


def test_shard_range_partitions_every_unit_once():
    for n in (0, 1, 7, 8, 64, 1000):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                f, c = sharding.shard_range(n, world, r)
                seen.extend(range(f, f + c))
                for u in range(f, f + c):
                    assert sharding.unit_owner(u, n, world) == r
            assert seen == list(range(n))
            counts = [sharding.shard_range(n, world, r)[1] for r in range(world)]
            assert max(counts) - min(counts) <= 1


def test_shard_range_rejects_bad_arguments():
    with pytest.raises(ValueError):
        sharding.shard_range(4, 0, 0)
    with pytest.raises(ValueError):
        sharding.shard_range(4, 2, 2)
    with pytest.raises(ValueError):
        sharding.unit_owner(4, 4, 2)


def _demod_bits(x):
    r = orc.demod_gmsk(x, sps=1, filter_width=20000)
    return r["bits_a"], r["bits_b"]


def _worker(rank, world, port, iq, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        first, count = sharding.shard_range(iq.shape[0], world, rank)
        local = [_demod_bits(iq[b]) for b in range(first, first + count)]
        full = sharding.gather_units(local, iq.shape[0])
        # whole-job sample count the way bench.py aggregates it
        t = torch.tensor([count * iq.shape[1]], dtype=torch.int64)
        dist.all_reduce(t)
        if rank == 0:
            q.put((int(t.item()), [(a.tobytes(), b.tobytes()) for a, b in full]))
        else:
            assert full is None
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_shard_and_gather_matches_single_process():
    B = 5  # odd on purpose: ranks own 3 and 2 streams
    iq = sig.make_batch("gmsk10k", B, nframes=2, device_rate=1000000, seed=21)
    want = [tuple(x.tobytes() for x in _demod_bits(iq[b])) for b in range(B)]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, iq, q)) for r in range(2)]
    for p in procs:
        p.start()
    total, got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert total == B * iq.shape[1]
    assert got == want


# ---- channel-sharded multi-carrier receiver (C4, SURVEY 8e PFB form): channelize locally, all-to-all the channel streams, per-channel
# chains on the owner.  CPU stand-in: the oracle's channelizer / per-channel chains around the SAME exchange bench.py uses: the C++
# chan_exchange::all_to_all of qradiolink_amd/host/chan_cluster.cpp (libqrl_cluster.so), here with its callback transport around gloo
# (sharding.exchange_channels -> Exchange.torch); bench.py --config c4 --gpus N drives the same object with the RCCL transport.
def _c4_worker(rank, world, port, iq, M, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        B = iq.shape[0]
        first, count = sharding.shard_range(B, world, rank)
        assert count == B // world
        taps = orc.chan_proto_taps(M)
        cpr = M // world                                            # channels per rank
        chans = np.stack([orc.pfb_channelizer(iq[b], taps, M) for b in range(first, first + count)])   # [B_local, M, n1]
        n1 = chans.shape[2]
        send = torch.from_numpy(np.ascontiguousarray(chans.reshape(count, world, cpr, n1).transpose(1, 0, 2, 3)))   # [dest, B_local, cpr, n1]
        recv = sharding.exchange_channels(send).numpy()           # [src, B_local, cpr, n1] = stream-major, this rank's channels
        mine = recv.reshape(B * cpr, n1)                           # row = stream * cpr + local channel
        out16, rssi, dib = orc.mmdvm_channel_tails(mine, cal=1.0)
        res = [(out16[r].tobytes(), rssi[r].tobytes(), dib[r].tobytes()) for r in range(B * cpr)]
        box = [None] * world if rank == 0 else None
        dist.gather_object(res, box, dst=0)
        if rank == 0:
            q.put((box, sharding.bytes_per_link_per_step(count, M, world, n1)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_channel_all_to_all_matches_single_process():
    M, B, n = 10, 4, 10 * 3000
    rng = np.random.default_rng(3)
    t = np.arange(n)
    iq = (0.02 * (rng.standard_normal((B, n)) + 1j * rng.standard_normal((B, n)))).astype(np.complex64)
    for b in range(B):
        for c in (1, 4, 8):
            f0 = c * 25000.0 if c <= M // 2 else (c - M) * 25000.0
            iq[b] += (0.2 * np.exp(1j * (2 * np.pi * f0 * t / (25000.0 * M) + 2.0 * np.sin(2 * np.pi * (300 + 50 * b + 10 * c) * t / (25000.0 * M))))).astype(np.complex64)
    want = []
    for b in range(B):
        o, r = orc.demod_mmdvm_multi_rssi(iq[b], M, cal=1.0)
        _, d = orc.demod_mmdvm_multi_4fsk(iq[b], M)
        want.append([(o[c].tobytes(), r[c].tobytes(), d[c].tobytes()) for c in range(M)])
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 2
    procs = [ctx.Process(target=_c4_worker, args=(r, world, port, iq, M, q)) for r in range(world)]
    for p in procs:
        p.start()
    box, link_bytes = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cpr = M // world
    for r in range(world):                 # rank r owns channels [r cpr, (r + 1) cpr) of every stream
        for b in range(B):
            for cl in range(cpr):
                assert box[r][b * cpr + cl] == want[b][r * cpr + cl], (r, b, cl)
    assert link_bytes == (B // world) * cpr * (n // M) * 8
