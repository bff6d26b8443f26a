"""Multi-GPU sharding of the RX hot path: the path shards by independent unit
(streams for the single-carrier chains, channels for the MMDVM channelizer) with NO data-path
collective: every rank runs the same kernel pipeline on its own units.  torch.distributed (RCCL on
the GPU box, gloo in the CPU tests) is used only to gather the small per-unit results (decoded bits /
counts) on rank 0, the way gr_mmdvm_sink gives every channel its own socket
(reference src/gr/gr_mmdvm_sink.cpp:77-173).

The one real exchange step is the channel-sharded multi-carrier receiver (C4, SURVEY.md 8e "PFB form"): every rank channelizes ITS
wideband inputs, exchange_channels() (one all_to_all_single: RCCL over xGMI on the GPU box) hands each channel's 25 ksps samples to
the rank that owns the channel, and that rank runs the per-channel chains.  Per link and step this moves 1 / world of a rank's
channel samples (= 1 / world of its input bytes), against the whole wideband batch for an input broadcast."""


def shard_range(n_units, world, rank):
    """Balanced contiguous split of n_units over `world` ranks: returns (first, count) for `rank`.
    The first n_units % world ranks get one extra unit."""
    if world < 1 or not (0 <= rank < world) or n_units < 0:
        raise ValueError("bad shard request: n_units=%r world=%r rank=%r" % (n_units, world, rank))
    base, extra = divmod(n_units, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def unit_owner(unit, n_units, world):
    """Rank that owns `unit` under shard_range (inverse mapping)."""
    if not (0 <= unit < n_units):
        raise ValueError("unit out of range")
    base, extra = divmod(n_units, world)
    edge = extra * (base + 1)
    if unit < edge:
        return unit // (base + 1)
    return extra + (unit - edge) // base


def gather_units(local_items, n_units, dist=None):
    """Gather per-unit python/numpy results from all ranks on rank 0 in global unit order.
    local_items: list with one entry per locally owned unit (shard_range order).
    Returns the full list on rank 0 and None elsewhere.  Without an initialised process group the
    local list is returned (world 1)."""
    if dist is None:
        import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        if len(local_items) != n_units:
            raise ValueError("single process must own every unit")
        return list(local_items)
    world, rank = dist.get_world_size(), dist.get_rank()
    first, count = shard_range(n_units, world, rank)
    if len(local_items) != count:
        raise ValueError("rank %d owns %d units, got %d results" % (rank, count, len(local_items)))
    box = [None] * world if rank == 0 else None
    dist.gather_object(list(local_items), box, dst=0)
    if rank != 0:
        return None
    out = []
    for r in range(world):
        f, c = shard_range(n_units, world, r)
        if len(box[r]) != c:
            raise RuntimeError("rank %d returned %d units, expected %d" % (r, len(box[r]), c))
        out.extend(box[r])
    return out


# ---- the exchange itself lives in C++ (qradiolink_amd/host/chan_cluster.*, libqrl_cluster.so): chan_exchange::all_to_all with the
# transports rccl (production, N GPUs), self (one rank) and callback (a function: torch.distributed / gloo in the CPU tests, a
# permutation copy in the single-device emulation).  bench.py --config c4 --gpus N, tests/test_sharding.py and
# tests/test_gpu_sharding.py all call Exchange.all_to_all / Cluster.step below: one code path.
import ctypes as _C
import os as _os

_EXCHANGE_FN = _C.CFUNCTYPE(_C.c_int, _C.c_void_p, _C.c_void_p, _C.c_void_p, _C.c_size_t, _C.c_void_p)
_cluster_lib = None


def cluster_library():
    global _cluster_lib
    if _cluster_lib is None:
        path = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "libqrl_cluster.so")
        if not _os.path.exists(path):
            raise RuntimeError("libqrl_cluster.so is not built: make -C qradiolink_amd/csrc cluster")
        L = _C.CDLL(path)
        vp, sz = _C.c_void_p, _C.c_size_t
        L.qrl_cluster_last_error.restype = _C.c_char_p
        L.qrl_exchange_unique_id.argtypes = [vp]
        L.qrl_exchange_create_rccl.argtypes = [_C.c_int, _C.c_int, vp, _C.POINTER(vp)]
        L.qrl_exchange_create_self.argtypes = [_C.POINTER(vp)]
        L.qrl_exchange_create_callback.argtypes = [_C.c_int, _C.c_int, _EXCHANGE_FN, vp, _C.POINTER(vp)]
        L.qrl_exchange_all_to_all.argtypes = [vp, vp, vp, sz, vp]
        L.qrl_exchange_destroy.argtypes = [vp]
        L.qrl_cluster_create.argtypes = [vp, vp, _C.c_int, _C.c_int, sz, _C.POINTER(vp)]
        L.qrl_cluster_destroy.argtypes = [vp]
        for n in ("qrl_cluster_front", "qrl_cluster_tail"):
            getattr(L, n).restype = vp
            getattr(L, n).argtypes = [vp]
        L.qrl_cluster_rows.argtypes = [vp]
        L.qrl_cluster_step.argtypes = [vp, vp, sz, sz, vp, sz, vp]
        L.qrl_cluster_channelize.argtypes = [vp, vp, sz, sz]
        L.qrl_cluster_exchange.argtypes = [vp]
        L.qrl_cluster_exchange_begin.argtypes = [vp]
        L.qrl_cluster_exchange_end.argtypes = [vp]
        L.qrl_exchange_group_create.argtypes = [_C.c_int, _C.POINTER(vp)]
        L.qrl_exchange_group_member.argtypes = [vp, _C.c_int, _C.POINTER(vp)]
        L.qrl_exchange_group_bytes_moved.argtypes = [vp]
        L.qrl_exchange_group_bytes_moved.restype = _C.c_ulonglong
        L.qrl_exchange_group_skip_copies.argtypes = [vp, _C.c_int]
        L.qrl_exchange_group_destroy.argtypes = [vp]
        L.qrl_cluster_process_channels.argtypes = [vp, vp, sz, vp]
        L.qrl_cluster_sync.argtypes = [vp]
        _cluster_lib = L
    return _cluster_lib


def _ck(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, cluster_library().qrl_cluster_last_error().decode()))


class Exchange:
    """qrl_host::chan_exchange behind its C ABI.  Exchange.rccl(dist): ncclAllToAll on a communicator of this job's ranks (the unique id
    travels through the given torch.distributed group); Exchange.torch(dist): the callback transport around
    torch.distributed.all_to_all_single (gloo on CPU tensors in the tests); Exchange.callback(world, rank, fn): any function
    fn(send_ptr, recv_ptr, bytes_per_peer, stream) -> None; Exchange.self_(): one rank, a device copy."""

    def __init__(self, handle, world, rank, keep=None):
        self.h, self.world, self.rank, self._keep = handle, world, rank, keep
        self.tensors = {}

    @classmethod
    def self_(cls):
        h = _C.c_void_p()
        _ck(cluster_library().qrl_exchange_create_self(_C.byref(h)), "qrl_exchange_create_self")
        return cls(h, 1, 0)

    @classmethod
    def callback(cls, world, rank, fn):
        def thunk(user, send, recv, nbytes, stream):
            try:
                fn(send, recv, nbytes, stream)
                return 0
            except Exception as e:          # a Python exception must not unwind through the C++ frames
                import sys
                print("exchange callback failed: %r" % (e,), file=sys.stderr)
                return 1
        cfn = _EXCHANGE_FN(thunk)
        h = _C.c_void_p()
        _ck(cluster_library().qrl_exchange_create_callback(world, rank, cfn, None, _C.byref(h)), "qrl_exchange_create_callback")
        return cls(h, world, rank, keep=cfn)

    @classmethod
    def torch(cls, dist=None):
        """all_to_all_single of the process group, on the tensors registered for the pointers the C++ side passes down"""
        if dist is None:
            import torch.distributed as dist
        world, rank = dist.get_world_size(), dist.get_rank()
        ex = None

        def fn(send, recv, nbytes, stream):
            s, r = ex.tensors[send], ex.tensors[recv]
            dist.all_to_all_single(r, s)
        ex = cls.callback(world, rank, fn)
        return ex

    @classmethod
    def rccl(cls, dist=None):
        """the production transport: an RCCL communicator of this job's ranks (one per GPU); the 128-byte unique id is made by rank 0
        and broadcast through the existing torch.distributed group"""
        L = cluster_library()
        single = dist is None or not (dist.is_available() and dist.is_initialized())     # no process group: a communicator of one rank
        world, rank = (1, 0) if single else (dist.get_world_size(), dist.get_rank())
        buf = (_C.c_ubyte * 128)()
        if rank == 0:
            _ck(L.qrl_exchange_unique_id(buf), "qrl_exchange_unique_id")
        box = [bytes(buf)]
        if not single:
            dist.broadcast_object_list(box, src=0)
        ident = (_C.c_ubyte * 128).from_buffer_copy(box[0])
        h = _C.c_void_p()
        _ck(L.qrl_exchange_create_rccl(world, rank, ident, _C.byref(h)), "qrl_exchange_create_rccl")
        return cls(h, world, rank)

    def register(self, *tensors):
        """the callback transports look their tensors up by data pointer"""
        for t in tensors:
            self.tensors[t.data_ptr()] = t

    def all_to_all(self, send, recv, stream=0):
        """send / recv: contiguous tensors of `world` equal blocks (block d goes to rank d, block s came from rank s)"""
        assert send.is_contiguous() and recv.is_contiguous() and send.numel() == recv.numel() and send.shape[0] == self.world
        self.register(send, recv)
        nbytes = send.numel() * send.element_size() // self.world
        _ck(cluster_library().qrl_exchange_all_to_all(self.h, send.data_ptr(), recv.data_ptr(), nbytes, _C.c_void_p(stream)), "qrl_exchange_all_to_all")
        del self.tensors[send.data_ptr()], self.tensors[recv.data_ptr()]

    def close(self):
        if self.h:
            cluster_library().qrl_exchange_destroy(self.h)
            self.h = _C.c_void_p()


class LocalGroup:
    """qrl_host::local_group: N ranks of THIS process on one device -- a real all-to-all among the members' buffers (device copies, an all-to-all's
    dependency structure).  .member(rank) is that rank's Exchange.  The emulation loop: every rank channelize(), every rank exchange_begin(), every rank
    exchange_end(), every rank process_channels() (EmulatedRanks.step does it)."""

    def __init__(self, world):
        self.L, self.world = cluster_library(), world
        self.h = _C.c_void_p()
        _ck(self.L.qrl_exchange_group_create(world, _C.byref(self.h)), "qrl_exchange_group_create")

    def member(self, rank):
        h = _C.c_void_p()
        _ck(self.L.qrl_exchange_group_member(self.h, rank, _C.byref(h)), "qrl_exchange_group_member")
        return Exchange(h, self.world, rank)

    def bytes_moved(self):
        return int(self.L.qrl_exchange_group_bytes_moved(self.h))

    def skip_copies(self, skip):
        _ck(self.L.qrl_exchange_group_skip_copies(self.h, 1 if skip else 0), "qrl_exchange_group_skip_copies")

    def close(self):
        if self.h:
            self.L.qrl_exchange_group_destroy(self.h)
            self.h = _C.c_void_p()


class EmulatedRanks:
    """`world` chan_cluster objects of one process on one device behind a LocalGroup: the multi-GPU step of C4 at the N-rank SHAPE (streams_local
    wideband streams and num_channels / world channels per rank) on a one-GPU box.  step(iq): iq [world * streams_local, n] (rank r takes rows
    r * streams_local ...); nothing synchronises with the host."""

    def __init__(self, ctx, world, num_channels, streams_local, max_chunk):
        self.world, self.bl = world, streams_local
        self.group = LocalGroup(world)
        self.exs = [self.group.member(r) for r in range(world)]
        self.cls = [Cluster(ctx, self.exs[r], num_channels, streams_local, max_chunk) for r in range(world)]

    def step(self, iq):
        for r, c in enumerate(self.cls):
            c.channelize(iq[r * self.bl:(r + 1) * self.bl])
        for c in self.cls:
            c.exchange_begin()
        for c in self.cls:
            c.exchange_end()
        for c in self.cls:
            c.process_channels()

    def sync(self):
        for c in self.cls:
            c.sync()

    def close(self):
        for c in self.cls:
            c.close()
        for e in self.exs:
            e.close()
        self.group.close()


class Cluster:
    """qrl_host::chan_cluster: one rank of the channel-sharded C4 receiver (channelize own streams -> one all-to-all -> per-channel
    chains of the owned channels).  .front / .tail are Channelizer views of its two handles (profiling on .front; the int16 / RSSI /
    4FSK outputs live on .tail, rows = (source rank * streams_local + stream) * channels_per_rank + local channel)."""

    def __init__(self, ctx, exchange, num_channels, streams_local, max_chunk):
        import qradiolink_amd as q
        self.L, self.ex, self.ctx = cluster_library(), exchange, ctx
        self.h = _C.c_void_p()
        _ck(self.L.qrl_cluster_create(ctx.h, exchange.h, num_channels, streams_local, max_chunk, _C.byref(self.h)), "qrl_cluster_create")
        self.rows = self.L.qrl_cluster_rows(self.h)
        self.per = num_channels // exchange.world
        self.front = q.Channelizer(ctx, num_channels, batch=streams_local, max_chunk=max_chunk, _handle=self.L.qrl_cluster_front(self.h))
        self.tail = q.Channelizer(ctx, 1, batch=self.rows, max_chunk=max_chunk // num_channels, form=3, _handle=self.L.qrl_cluster_tail(self.h))

    def step_async(self, iq):
        t = self.tail
        _ck(self.L.qrl_cluster_step(self.h, iq.data_ptr(), iq.stride(0), iq.shape[1], t.out.data_ptr(), t.cap, t.counts.data_ptr()), "qrl_cluster_step")

    def channelize(self, iq):
        _ck(self.L.qrl_cluster_channelize(self.h, iq.data_ptr(), iq.stride(0), iq.shape[1]), "qrl_cluster_channelize")

    def exchange(self):
        _ck(self.L.qrl_cluster_exchange(self.h), "qrl_cluster_exchange")

    def exchange_begin(self):
        _ck(self.L.qrl_cluster_exchange_begin(self.h), "qrl_cluster_exchange_begin")

    def exchange_end(self):
        _ck(self.L.qrl_cluster_exchange_end(self.h), "qrl_cluster_exchange_end")

    def process_channels(self):
        t = self.tail
        _ck(self.L.qrl_cluster_process_channels(self.h, t.out.data_ptr(), t.cap, t.counts.data_ptr()), "qrl_cluster_process_channels")

    def sync(self):
        _ck(self.L.qrl_cluster_sync(self.h), "qrl_cluster_sync")

    def close(self):
        if self.h:
            self.front.close()
            self.tail.close()
            self.L.qrl_cluster_destroy(self.h)
            self.h = _C.c_void_p()


def exchange_channels(send, dist=None, exchange=None):
    """All-to-all of channelizer output through the C++ exchange (chan_exchange::all_to_all).  send: tensor [world, B_local, C // world,
    n1] (any device) whose slice [d] holds this rank's streams x the channels rank d owns.  Returns recv of the same shape with
    recv[s] = the slice rank s sent here, i.e. this rank's channels of every stream, stream-major by source rank.
    exchange: an Exchange to use (default: Exchange.torch(dist), the callback transport around the process group)."""
    import torch
    if dist is None:
        import torch.distributed as dist
    if exchange is None and (not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1):
        return send.clone()
    own = exchange is None
    ex = Exchange.torch(dist) if own else exchange
    if send.shape[0] != ex.world:
        raise ValueError("exchange_channels: leading dimension must be the world size")
    sv = (torch.view_as_real(send) if send.is_complex() else send).contiguous()
    recv = torch.empty_like(sv)
    ex.all_to_all(sv, recv)
    if own:
        ex.close()
    return torch.view_as_complex(recv) if send.is_complex() else recv


def bytes_per_link_per_step(b_local, channels, world, n1, itemsize=8):
    """what one rank sends to ONE peer per step in the all-to-all form: its streams x the peer's channels x n1 cf32 items"""
    return b_local * (channels // world) * n1 * itemsize
