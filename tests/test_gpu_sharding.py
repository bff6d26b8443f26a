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


def _unsharded(qrl_ctx, d, M, cuts, cal):
    """the unsharded receiver over the same calls: per call (int16, counts, rssi, rssi counts, dibits, 4fsk counts)"""
    import qradiolink_amd as q
    ref = q.Channelizer(qrl_ctx, M, batch=d.shape[0], max_chunk=max(cuts))
    ref.calibrate_rssi(cal)
    ref.enable_4fsk()
    out, pos = [], 0
    for c in cuts:
        ro, rc = ref.process(d[:, pos:pos + c].contiguous())
        pos += c
        out.append((ro.cpu().numpy().copy(), rc.cpu().numpy().copy(), ref.rssi.cpu().numpy().copy(), ref.rssi_counts.cpu().numpy().copy(),
                    ref.dibits.cpu().numpy().copy(), ref.fsk_counts.cpu().numpy().copy()))
    ref.close()
    return out


def _compare_rank(tail, ref_call, r, world, B, cpr):
    """rows of rank r's per-channel handle against the unsharded receiver: row = (source rank * Bl + stream) * cpr + local channel"""
    ro, rc, rr, rrc, rd, rdc = ref_call
    o, cn = tail.out.cpu().numpy(), tail.counts.cpu().numpy()
    rs, rsc = tail.rssi.cpu().numpy(), tail.rssi_counts.cpu().numpy()
    db, dbc = tail.dibits.cpu().numpy(), tail.fsk_counts.cpu().numpy()
    for b in range(B):                       # global stream index = source rank * Bl + local stream: rows are in that order
        for cl in range(cpr):
            row, ch_abs = b * cpr + cl, r * cpr + cl
            assert cn[row, 0] == rc[b, ch_abs] and np.array_equal(o[row, 0, :cn[row, 0]], ro[b, ch_abs, :rc[b, ch_abs]]), (r, b, cl)
            assert rsc[row, 0] == rrc[b, ch_abs] and np.array_equal(rs[row, 0, :rsc[row, 0]], rr[b, ch_abs, :rrc[b, ch_abs]])
            assert dbc[row, 0, 2] == rdc[b, ch_abs, 2] and np.array_equal(db[row, 0, :dbc[row, 0, 2]], rd[b, ch_abs, :rdc[b, ch_abs, 2]])


def test_channel_all_to_all_form_two_emulated_ranks_equal_the_unsharded_receiver(qrl_ctx):
    """SURVEY 8e, PFB form, through the SAME C++ object bench.py --config c4 --gpus N drives (qrl_host::chan_cluster, libqrl_cluster.so):
    every rank channelizes ITS wideband streams, chan_exchange::all_to_all hands each rank its channels of EVERY stream, the
    per-channel chains run there.  Two ranks emulated on one device: two clusters whose callback transport records the send / receive
    buffers; the test then performs the permutation an all-to-all performs (block r of rank s's send buffer -> block s of rank r's
    receive buffer) on each rank's exchange stream.  Two calls, so that channelizer history, channel-ring history and the symbol-sync
    state carry.  Must equal the unsharded receiver bit for bit."""
    import ctypes as C
    import torch
    import test_gpu_chan as tg
    M, B, world, n = 64, 4, 2, 64 * 1600
    cpr, Bl = M // world, B // world
    iq = tg._wideband(M, n, seed=91, nstreams=B)
    d = torch.from_numpy(iq).cuda()
    cuts = [64 * 1000, 64 * 600]
    ref = _unsharded(qrl_ctx, d, M, cuts, 0.25)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    pending = {}
    exs = [sharding.Exchange.callback(world, r, (lambda r: lambda send, recv, nbytes, stream: pending.__setitem__(r, (send, recv, nbytes, stream)))(r))
           for r in range(world)]
    cls = [sharding.Cluster(qrl_ctx, exs[r], M, Bl, max(cuts)) for r in range(world)]
    for c in cls:
        c.tail.calibrate_rssi(0.25)
        c.tail.enable_4fsk()
    pos = 0
    for k, c in enumerate(cuts):
        part = d[:, pos:pos + c].contiguous()
        pos += c
        for r in range(world):
            cls[r].channelize(part[r * Bl:(r + 1) * Bl].contiguous())
        for r in range(world):
            cls[r].exchange()                                               # the callbacks fill `pending`
        for r in range(world):
            cls[r].front.sync()                                             # (the emulation copies across ranks: every channelizer must be through)
        for r in range(world):
            _, recv, nbytes, stream = pending[r]
            for s_ in range(world):
                assert hip.hipMemcpyAsync(recv + s_ * nbytes, pending[s_][0] + r * nbytes, nbytes, 3, stream) == 0   # hipMemcpyDeviceToDevice
        for r in range(world):
            cls[r].process_channels()
            cls[r].sync()
            _compare_rank(cls[r].tail, ref[k], r, world, B, cpr)
    assert np.abs(ref[-1][0]).max() > 1000
    for c in cls:
        c.close()
    for e in exs:
        e.close()


def test_cluster_with_the_rccl_transport_on_one_rank_equals_the_plain_receiver(qrl_ctx):
    """the production transport on what one GPU allows: an RCCL communicator of ONE rank (ncclCommInitRank(1, id, 0)), chan_cluster::step
    = channelize -> ncclAllToAll (to itself) -> per-channel chains, two pipelined steps without a host synchronisation in between;
    rows equal the unsharded receiver's channels bit for bit.  Also the self transport (a device copy)."""
    import ctypes as C
    import torch
    import test_gpu_chan as tg
    M, B, n = 64, 3, 64 * 1200
    iq = tg._wideband(M, n, seed=92, nstreams=B)
    d = torch.from_numpy(iq).cuda()
    cuts = [64 * 700, 64 * 500]
    ref = _unsharded(qrl_ctx, d, M, cuts, -1.5)
    L = sharding.cluster_library()
    for kind in ("rccl", "self"):
        if kind == "rccl":
            ident = (C.c_ubyte * 128)()
            assert L.qrl_exchange_unique_id(ident) == 0, L.qrl_cluster_last_error()
            h = C.c_void_p()
            assert L.qrl_exchange_create_rccl(1, 0, ident, C.byref(h)) == 0, L.qrl_cluster_last_error()
            ex = sharding.Exchange(h, 1, 0)
        else:
            ex = sharding.Exchange.self_()
        cl = sharding.Cluster(qrl_ctx, ex, M, B, max(cuts))
        cl.tail.calibrate_rssi(-1.5)
        cl.tail.enable_4fsk()
        pos = 0
        for k, c in enumerate(cuts):
            cl.step_async(d[:, pos:pos + c].contiguous())
            pos += c
            cl.sync()
            _compare_rank(cl.tail, ref[k], 0, 1, B, M)
        cl.close()
        ex.close()


@pytest.mark.parametrize("world,Bl", [(8, 1), (4, 2)])
def test_emulated_ranks_with_the_local_group_transport_equal_the_unsharded_receiver(qrl_ctx, world, Bl):
    """the multi-GPU step at the 8-rank SHAPE on one device.  `world` chan_cluster objects of one process behind qrl_host::local_group -- a
    real all-to-all among their buffers (device copies with an all-to-all's dependencies: every receive side waits for every send buffer, every send
    buffer is released by its last reader) -- driven by sharding.EmulatedRanks.step with NO host synchronisation inside a step or between the three
    steps (the 3-slot pipeline: buffers are reused from the fourth step on, so five steps run).  Every rank's rows equal the unsharded receiver's
    channels bit for bit, step by step."""
    import torch
    import test_gpu_chan as tg
    M = 64
    B, cpr = world * Bl, M // world
    cuts = [64 * 700, 64 * 500, 64 * 700, 64 * 300, 64 * 600]
    iq = tg._wideband(M, sum(cuts), seed=93, nstreams=B)
    d = torch.from_numpy(iq).cuda()
    ref = _unsharded(qrl_ctx, d, M, cuts, 0.5)
    em = sharding.EmulatedRanks(qrl_ctx, world, M, Bl, max(cuts))
    for c in em.cls:
        c.tail.calibrate_rssi(0.5)
        c.tail.enable_4fsk()
    # the outputs of a step live in the per-channel handle's own buffers, which the next step overwrites: snapshot them on the handle's stream order by
    # synchronising only AFTER queueing the whole step (the step itself has no host synchronisation)
    pos = 0
    for k, c in enumerate(cuts):
        em.step(d[:, pos:pos + c].contiguous())
        pos += c
        em.sync()
        for r in range(world):
            _compare_rank(em.cls[r].tail, ref[k], r, world, B, cpr)
    assert em.group.bytes_moved() == sum(world * world * Bl * cpr * max(cuts) // 64 * 8 for _ in cuts)
    em.close()
