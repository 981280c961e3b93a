"""Multi-GPU sharding of the hot path: projection angles split contiguously over the ranks.

Every ray and every detector pixel is independent, so the only exchange is assembling the
sinogram: one all-gather (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests), plus
one scalar all-reduce(max) for the air mask of get_basismat_sinos (matdecomp.py:195-196 uses the
GLOBAL maximum of sinogram 1).  The phantom and the tables are replicated.
"""
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def split(n, rank, world_size):
    """Contiguous block of rank: the first n % world ranks get one extra item."""
    base, rem = divmod(n, world_size)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def my_views(n_views):
    r, w = world()
    return split(n_views, r, w)


def _needs_cpu_staging(t):
    """gloo (CPU tests, single-GPU rehearsals) cannot move device tensors in every collective."""
    return t.is_cuda and dist.get_backend() == 'gloo'


# Buffers of the gather, allocated once per (caller tag, shape, dtype, device) and reused by every later call with the
# same tag: the padded send and receive buffers of ragged shards / the gloo rehearsal, and - only for callers that ask
# for it with ``reuse_out=True`` - the result tensor.
_buffers = {}


def _buffer(kind, shape, dtype, device, pin=False, tag=None):
    key = (tag, kind, tuple(shape), dtype, str(device))
    t = _buffers.get(key)
    if t is None:
        t = torch.empty(tuple(shape), dtype=dtype, device=device, pin_memory=bool(pin and str(device) == 'cpu' and torch.cuda.is_available()))
        _buffers[key] = t
    return t


def release_buffers():
    """Drop the cached gather buffers (they are sized like the full sinogram)."""
    _buffers.clear()


def gather_views(local, n_views, view_dim=0, async_op=False, out=None, tag=None, reuse_out=False):
    """All-gather view shards (possibly of unequal size) into the full tensor on every rank.

    ``local`` is this rank's contiguous shard with the views on ``view_dim`` (0, or 1 with a leading "spectrum"
    dimension: ``[S, views, ...]``, the layout the projection writes).  The shards go straight from that buffer into
    the result, one collective per leading index (each a contiguous block of views on both sides): no transposed copy,
    no concatenation.  The result is a NEW tensor owned by the caller unless ``out=`` is given or ``reuse_out=True``
    (then it lives in a buffer cached under ``tag`` and is overwritten by the next such gather of the same shape: for
    step loops that consume the result at once, like bench.py - zero device allocations per call).
    Ragged shards (n_views not a multiple of the world size) are padded through preallocated send / receive buffers
    cached under ``tag`` and compacted with in-place copies: two gathers of the same shape that may be in flight at the
    same time must use different tags.  With ``async_op=True`` returns ``finish`` - call it to wait and get the tensor
    (on the nccl/RCCL backend the transfers run on the process group's own stream and overlap the caller's kernels)."""
    r, w = world()
    if w == 1:
        return (lambda: local) if async_op else local
    if view_dim not in (0, 1) or view_dim >= local.dim():
        raise ValueError('views must be dimension 0, or 1 behind one leading dimension')
    if not local.is_contiguous():
        local = local.contiguous()
    lead = local.shape[0] if view_dim == 1 else 1
    tail = tuple(local.shape[view_dim + 1:])
    sizes = [split(n_views, k, w) for k in range(w)]
    n_mine = sizes[r][1] - sizes[r][0]
    if local.shape[view_dim] != n_mine:
        raise ValueError(f'rank {r} holds {local.shape[view_dim]} views, its share of {n_views} is {n_mine}')
    n_max = max(e - b for b, e in sizes)
    ragged = n_max * w != n_views
    dev = local.device
    staged = _needs_cpu_staging(local)                     # gloo rehearsal: collectives on host copies
    cdev = torch.device('cpu') if staged else dev
    full_shape = ((lead,) if view_dim == 1 else ()) + (n_views,) + tail
    if out is None:
        out = (_buffer('out', full_shape, local.dtype, dev, tag=tag) if reuse_out else
               torch.empty(full_shape, dtype=local.dtype, device=dev))
    elif tuple(out.shape) != full_shape or not out.is_contiguous():
        raise ValueError(f'out must be a contiguous tensor of shape {full_shape}')
    loc3 = local.view((lead, n_mine) + tail)
    out3 = out.view((lead, n_views) + tail)
    direct = not ragged and not staged                     # shards land in the result as they arrive
    works, recvs = [], []
    for s in range(lead):
        if direct:
            src, dst = loc3[s], out3[s]
        else:
            src = _buffer(('send', s), (n_max,) + tail, local.dtype, cdev, pin=True, tag=tag)
            src[:n_mine].copy_(loc3[s], non_blocking=not staged)
            dst = _buffer(('recv', s), (w * n_max,) + tail, local.dtype, cdev, pin=True, tag=tag)
        works.append(dist.all_gather_into_tensor(dst, src, async_op=async_op))
        recvs.append(dst)

    def finish():
        for wk in works:
            if async_op and wk is not None:
                wk.wait()
        if not direct:
            for s in range(lead):
                for k, (b, e) in enumerate(sizes):
                    out3[s, b:e].copy_(recvs[s][k * n_max:k * n_max + (e - b)], non_blocking=not staged)
        return out

    return finish if async_op else finish()


def global_max(value):
    """max over ranks of a 0-d tensor (device or CPU).  NaN-propagating like np.max (matdecomp.py:195-196): the
    collective's MAX does not define what a NaN does, so the NaN flag travels as a second element."""
    r, w = world()
    if w > 1:
        nan = torch.isnan(value)
        pair = torch.stack([torch.where(nan, torch.full_like(value, float('-inf')), value), nan.to(value.dtype)])
        if _needs_cpu_staging(pair):
            v = pair.cpu()
            dist.all_reduce(v, op=dist.ReduceOp.MAX)
            pair = v.to(value.device)
        else:
            dist.all_reduce(pair, op=dist.ReduceOp.MAX)
        value.copy_(torch.where(pair[1] > 0, torch.full_like(value, float('nan')), pair[0]))
    return value
