"""Multi-GPU sharding of the RX hot path: the path shards by independent unit
(streams for the single-carrier chains, channels for the MMDVM channelizer) with NO data-path
collective: every rank runs the same kernel pipeline on its own units.  torch.distributed (RCCL on
the GPU box, gloo in the CPU tests) is used only to gather the small per-unit results (decoded bits /
counts) on rank 0, the way gr_mmdvm_sink gives every channel its own socket
(reference src/gr/gr_mmdvm_sink.cpp:77-173)."""


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
