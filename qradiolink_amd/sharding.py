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


def exchange_channels(send, dist=None):
    """All-to-all of channelizer output.  send: tensor [world, B_local, C // world, n1] (any device; complex64 viewed as float32 pairs is
    done here) whose slice [d] holds this rank's streams x the channels rank d owns.  Returns recv of the same shape with
    recv[s] = the slice rank s sent here, i.e. this rank's channels of every stream, stream-major by source rank."""
    import torch
    if dist is None:
        import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return send.clone()
    if send.shape[0] != dist.get_world_size():
        raise ValueError("exchange_channels: leading dimension must be the world size")
    sv = torch.view_as_real(send) if send.is_complex() else send
    recv = torch.empty_like(sv)
    dist.all_to_all_single(recv, sv.contiguous())
    return torch.view_as_complex(recv) if send.is_complex() else recv


def bytes_per_link_per_step(b_local, channels, world, n1, itemsize=8):
    """what one rank sends to ONE peer per step in the all-to-all form: its streams x the peer's channels x n1 cf32 items"""
    return b_local * (channels // world) * n1 * itemsize
