"""Multi-GPU paths exercised with the HIP pipeline itself.  A one-GPU box cannot run two RCCL ranks, so
  * stream sharding (C1-C3): two handles on one device take the two shards of sharding.shard_range; together they must equal
    the single-handle run and the oracle;
  * channel sharding (C4): TWO PROCESSES (gloo process group, both on cuda:0) -- rank 0 owns the wideband input and broadcasts it
    (bench.py does the same broadcast with RCCL on device tensors), every rank runs the HIP channelizer on ITS channel range,
    results are gathered with sharding.gather_units and compared with the oracle."""
import os
import socket

import numpy as np
import pytest

import orc
import sig
from qradiolink_amd import sharding

pytestmark = pytest.mark.gpu


def test_stream_sharding_two_handles_equal_single_handle(qrl_ctx):
    import torch
    import qradiolink_amd as q
    B, rate, offset = 5, 1000000, 1200.0
    iq = sig.make_batch("2fsk1k", B, nframes=2, device_rate=rate, rx_offset_hz=offset, seed=31)
    d = torch.from_numpy(iq).cuda()
    outs = []
    for rank in range(2):
        first, count = sharding.shard_range(B, 2, rank)
        dem = q.Demod(qrl_ctx, 18, batch=count, max_chunk=iq.shape[1], device_samp_rate=rate, carrier_offset_hz=offset)
        o = q.collect(dem, d[first:first + count].contiguous(), iq.shape[1])
        dem.close()
        outs.extend(zip(o["bits_a"], o["bits_b"], o["filtered"]))
    assert len(outs) == B
    for b in range(B):
        ref = orc.demod_2fsk(orc.frontend(iq[b], rate, offset), sps=10, filter_width=2000, fm=False)
        assert np.array_equal(outs[b][0], ref["bits_a"]) and np.array_equal(outs[b][1], ref["bits_b"])
        got, want = outs[b][2].view(np.float32) + np.float32(0), ref["filtered"].view(np.float32) + np.float32(0)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def _chan_worker(rank, world, port, iq_host, M, q_out):
    import torch
    import torch.distributed as dist
    import qradiolink_amd as q
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = iq_host.shape[1]
        buf = torch.from_numpy(iq_host.view(np.float32).copy()) if rank == 0 else torch.empty((iq_host.shape[0], 2 * n), dtype=torch.float32)
        dist.broadcast(buf, src=0)                       # the exchange step of SURVEY 8e (RCCL broadcast in bench.py --config c4)
        iq = torch.view_as_complex(buf.view(iq_host.shape[0], n, 2)).cuda()
        first, count = sharding.shard_range(M, world, rank)
        ctx = q.Context(0)
        ch = q.Channelizer(ctx, M, batch=iq.shape[0], max_chunk=n, channel_first=first, channel_count=count)
        out, cnt = ch.process(iq)
        out, cnt = out.cpu().numpy(), cnt.cpu().numpy()
        local = [out[0, c, :cnt[0, c]].copy() for c in range(count)]
        ch.close()
        ctx.close()
        full = sharding.gather_units(local, M)
        if rank == 0:
            q_out.put([x.tobytes() for x in full])
    finally:
        dist.destroy_process_group()


def test_channel_sharding_two_processes_with_broadcast():
    import torch.multiprocessing as mp
    M, n = 64, 64 * 1200
    rng = np.random.default_rng(4)
    iq = (0.05 * (rng.standard_normal((1, n)) + 1j * rng.standard_normal((1, n)))).astype(np.complex64)
    t = np.arange(n)
    for c in (3, 17, 40, 61):   # a few carriers so that the channels differ
        iq[0] += (0.2 * np.exp(2j * np.pi * (c / M + 0.0007) * t)).astype(np.complex64)
    ref = orc.demod_mmdvm_multi(iq[0], M)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q_out = ctx.Queue()
    procs = [ctx.Process(target=_chan_worker, args=(r, 2, port, iq, M, q_out)) for r in range(2)]
    for p in procs:
        p.start()
    got = q_out.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert len(got) == M
    for c in range(M):
        assert got[c] == np.ascontiguousarray(ref[c]).tobytes(), "channel %d differs from the oracle" % c


def test_channel_all_to_all_form_two_emulated_ranks_equal_the_unsharded_receiver(qrl_ctx):
    """SURVEY 8e, PFB form, as bench.py --config c4 --gpus N runs it: every rank channelizes ITS wideband streams
    (qrl_chan_channelize, output grouped by destination rank), an all-to-all hands each rank its channels of EVERY stream, the
    per-channel chains run there (form-3 handles, qrl_chan_process_channels).  Two ranks emulated on one device: the exchange is the
    same slicing all_to_all_single performs (sharding.exchange_channels; its gloo run is tests/test_sharding.py).  Two calls, so that
    channelizer history, channel-ring history and the symbol-sync state carry.  Must equal the unsharded receiver bit for bit."""
    import torch
    import qradiolink_amd as q
    import test_gpu_chan as tg
    M, B, world, n = 64, 4, 2, 64 * 1600
    cpr, Bl = M // world, B // world
    iq = tg._wideband(M, n, seed=91, nstreams=B)
    d = torch.from_numpy(iq).cuda()
    cuts = [64 * 1000, 64 * 600]
    # unsharded reference run
    ref = q.Channelizer(qrl_ctx, M, batch=B, max_chunk=max(cuts))
    ref.calibrate_rssi(0.25)
    ref.enable_4fsk()
    chans = [q.Channelizer(qrl_ctx, M, batch=Bl, max_chunk=max(cuts)) for _ in range(world)]
    tails = [q.Channelizer(qrl_ctx, 1, batch=B * cpr, max_chunk=max(cuts) // M, form=3) for _ in range(world)]
    for t in tails:
        t.calibrate_rssi(0.25)
        t.enable_4fsk()
    pos = 0
    for c in cuts:
        part = d[:, pos:pos + c].contiguous()
        pos += c
        n1 = c // M
        ro, rc = ref.process(part)
        ro, rc = ro.cpu().numpy(), rc.cpu().numpy()
        rr, rrc = ref.rssi.cpu().numpy(), ref.rssi_counts.cpu().numpy()
        rd, rdc = ref.dibits.cpu().numpy(), ref.fsk_counts.cpu().numpy()
        send = [torch.zeros((world, Bl, cpr, n1), dtype=torch.complex64, device="cuda") for _ in range(world)]
        ts = torch.cuda.current_stream().cuda_stream                          # the stream torch (and, in the real job, the collective) works on
        for r in range(world):
            chans[r].wait_for(ts)                                             # the send buffer was zero-filled on torch's stream
            chans[r].channelize_async(part[r * Bl:(r + 1) * Bl].contiguous(), send[r], world)
            chans[r].sync()
        for r in range(world):
            recv = torch.stack([send[s][r] for s in range(world)])            # what all_to_all_single delivers to rank r ...
            tails[r].wait_for(ts)                                             # ... on torch's stream: the handle's own stream has to wait for it
            tails[r].process_channels_async(recv.reshape(B * cpr, n1).contiguous(), n1)
            tails[r].sync()
            o, cn = tails[r].out.cpu().numpy(), tails[r].counts.cpu().numpy()
            rs, rsc = tails[r].rssi.cpu().numpy(), tails[r].rssi_counts.cpu().numpy()
            db, dbc = tails[r].dibits.cpu().numpy(), tails[r].fsk_counts.cpu().numpy()
            for b in range(B):
                for cl in range(cpr):
                    row, ch_abs = b * cpr + cl, r * cpr + cl
                    assert cn[row, 0] == rc[b, ch_abs] and np.array_equal(o[row, 0, :cn[row, 0]], ro[b, ch_abs, :rc[b, ch_abs]]), (r, b, cl)
                    assert rsc[row, 0] == rrc[b, ch_abs] and np.array_equal(rs[row, 0, :rsc[row, 0]], rr[b, ch_abs, :rrc[b, ch_abs]])
                    assert dbc[row, 0, 2] == rdc[b, ch_abs, 2] and np.array_equal(db[row, 0, :dbc[row, 0, 2]], rd[b, ch_abs, :rdc[b, ch_abs, 2]])
    assert np.abs(ro).max() > 1000
    for h in [ref] + chans + tails:
        h.close()
