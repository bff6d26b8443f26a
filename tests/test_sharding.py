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
