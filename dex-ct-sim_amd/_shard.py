"""Multi-GPU sharding of the hot path: projection angles split contiguously over the ranks.

Every ray and every detector pixel is independent, so the only exchange is assembling the
sinogram - a gather to one rank or to all of them, as point-to-point transfers over the direct xGMI
links or as one all-gather (RCCL when the backend is "nccl"; gloo in the CPU tests) -, plus
one scalar all-reduce(max) for the air mask of get_basismat_sinos (matdecomp.py:195-196 uses the
GLOBAL maximum of sinogram 1).  The phantom and the tables are replicated.
"""
import os

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


# How the shards travel (``mode=`` of gather_views; DEXCT_GATHER in the environment):
#   'root'    the gather of BASELINE.json's north star: every rank sends its shard to ``root`` (one point-to-point transfer per
#             peer, each over the direct xGMI link of its pair of GPUs, all in one RCCL group); only the root holds the sinogram
#             afterwards (the others get None).  7/8 of the sinogram cross the fabric once.
#   'direct'  every rank sends its shard to every peer and receives every peer's: an all-gather made of world-1 sends and
#             world-1 receives per rank, one per link - whatever algorithm RCCL would pick for ncclAllGather (a ring is bound by
#             ONE link: 1.5x at 8 GPUs predicted in profiles/r04c_shard_of.md) is not in the path.  Every rank holds the result.
#   'all'     one all_gather_into_tensor per leading index (rounds 1-4).
# The drop-in calls (get_sinos, get_basismat_sinos) return the assembled array to EVERY process, as the reference's call returns
# it to its one process: their default is 'direct'.  bench.py assembles on the root, as the north star says.
GATHER_MODES = ('root', 'direct', 'all')
DEFAULT_GATHER_MODE = os.environ.get('DEXCT_GATHER', 'direct')
if DEFAULT_GATHER_MODE not in GATHER_MODES:
    raise ValueError(f'DEXCT_GATHER={DEFAULT_GATHER_MODE!r}: one of {GATHER_MODES}')


def dropin_mode():
    """The mode of the drop-in calls (get_sinos, get_basismat_sinos): they return the assembled array to EVERY process, so a
    DEXCT_GATHER=root default - meant for step loops that assemble on one rank, like bench.py - cannot serve them (the other
    ranks would hold None and leave rank 0 alone in the next collective: advisor finding of round 5): 'all' when the
    environment asks for it, else 'direct'."""
    return 'all' if DEFAULT_GATHER_MODE == 'all' else 'direct'


def part_bounds(n_views, world_size, part):
    """[(begin, end) per rank] of piece ``part = (j, n_parts)`` of every rank's shard, in GLOBAL views: each shard is cut into
    n_parts contiguous pieces by the same rule (split), so every rank knows where every peer's piece j belongs."""
    j, n_parts = part
    out = []
    for k in range(world_size):
        b, e = split(n_views, k, world_size)
        pb, pe = split(e - b, j, n_parts)
        out.append((b + pb, b + pe))
    return out


def gather_views(local, n_views, view_dim=0, async_op=False, out=None, tag=None, reuse_out=False, mode=None, root=0, part=None):
    """Assemble view shards (possibly of unequal size) into the full tensor - on every rank ('direct', 'all') or on ``root``
    only ('root': the other ranks get None); see GATHER_MODES above.

    ``local`` is this rank's contiguous shard with the views on ``view_dim`` (0, or 1 with a leading "spectrum"
    dimension: ``[S, views, ...]``, the layout the projection writes).  The shards go straight from that buffer into
    the result (each a contiguous block of views on both sides, per leading index): no transposed copy, no concatenation.
    The result is a NEW tensor owned by the caller unless ``out=`` is given or ``reuse_out=True`` (then it lives in a buffer
    cached under ``tag`` and is overwritten by the next such gather of the same shape: for step loops that consume the
    result at once, like bench.py - zero device allocations per call).
    ``part=(j, n_parts)``: ``local`` is only piece j of this rank's shard (part_bounds) - a step that projects its views in
    chunks starts the transfer of a chunk as soon as it exists, while the next one is computed; all parts of one assembly
    must go to the same ``out`` (or the same ``tag`` with ``reuse_out``).
    Mode 'all' pads ragged shards through preallocated send / receive buffers cached under ``tag`` and compacts them with
    in-place copies (two gathers of the same shape that may be in flight at the same time must use different tags); the
    point-to-point modes need no padding.  With ``async_op=True`` returns ``finish`` - call it to wait and get the tensor (on
    the nccl/RCCL backend the transfers run on the process group's own stream and overlap the caller's kernels)."""
    r, w = world()
    mode = DEFAULT_GATHER_MODE if mode is None else mode
    if mode not in GATHER_MODES:
        raise ValueError(f'mode={mode!r}: one of {GATHER_MODES}')
    if w == 1:
        return (lambda: local) if async_op else local
    if view_dim not in (0, 1) or view_dim >= local.dim():
        raise ValueError('views must be dimension 0, or 1 behind one leading dimension')
    if not 0 <= root < w:
        raise ValueError(f'root={root} of {w} ranks')
    if part is not None and not (0 <= part[0] < part[1]):
        raise ValueError(f'part={part}')
    if not local.is_contiguous():
        local = local.contiguous()
    lead = local.shape[0] if view_dim == 1 else 1
    tail = tuple(local.shape[view_dim + 1:])
    sizes = [split(n_views, k, w) for k in range(w)] if part is None else part_bounds(n_views, w, part)
    n_mine = sizes[r][1] - sizes[r][0]
    if local.shape[view_dim] != n_mine:
        raise ValueError(f'rank {r} holds {local.shape[view_dim]} views, its share of {n_views}'
                         + (f' (piece {part[0]} of {part[1]})' if part else '') + f' is {n_mine}')
    dev = local.device
    staged = _needs_cpu_staging(local)                     # gloo rehearsal: transfers on host copies
    cdev = torch.device('cpu') if staged else dev
    full_shape = ((lead,) if view_dim == 1 else ()) + (n_views,) + tail
    receives = mode != 'root' or r == root
    if part is not None and out is None and not reuse_out:
        raise ValueError('the parts of one assembly need a common destination: out= or reuse_out=True with a tag')
    if not receives:
        out = None
    elif out is None:
        out = (_buffer('out', full_shape, local.dtype, dev, tag=tag) if reuse_out else
               torch.empty(full_shape, dtype=local.dtype, device=dev))
    elif tuple(out.shape) != full_shape or not out.is_contiguous():
        raise ValueError(f'out must be a contiguous tensor of shape {full_shape}')
    loc3 = local.view((lead, n_mine) + tail)
    out3 = out.view((lead, n_views) + tail) if receives else None
    pkey = None if part is None else tuple(part)

    if mode == 'all':
        n_max = max(e - b for b, e in sizes)
        ragged = n_max * w != n_views or part is not None
        direct = not ragged and not staged                 # shards land in the result as they arrive
        works, recvs = [], []
        for s in range(lead):
            if direct:
                src, dst = loc3[s], out3[s]
            else:
                src = _buffer(('send', s, pkey), (n_max,) + tail, local.dtype, cdev, pin=True, tag=tag)
                src[:n_mine].copy_(loc3[s], non_blocking=not staged)
                dst = _buffer(('recv', s, pkey), (w * n_max,) + tail, local.dtype, cdev, pin=True, tag=tag)
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

    # point-to-point: one send per (peer that receives, leading index), one receive per (peer, leading index), posted in the
    # same order on both sides; all of them in one batch (one RCCL group: they run concurrently, each pair over its own link)
    ops, landed = [], []
    if receives and n_mine:
        out3[:, sizes[r][0]:sizes[r][1]].copy_(loc3, non_blocking=True)      # this rank's own shard: a device copy
    for k in range(w):
        if k == r:
            continue
        if (mode == 'direct' or k == root) and n_mine:
            for s in range(lead):
                src = loc3[s]
                if staged:
                    src = _buffer(('p2p_send', s, k, pkey), src.shape, local.dtype, cdev, pin=True, tag=tag)
                    src.copy_(loc3[s])
                ops.append(dist.P2POp(dist.isend, src, k))
        b, e = sizes[k]
        if receives and e > b:
            for s in range(lead):
                dst = out3[s, b:e]
                if staged:
                    dst = _buffer(('p2p_recv', s, k, pkey), dst.shape, local.dtype, cdev, pin=True, tag=tag)
                    landed.append((s, b, e, dst))
                ops.append(dist.P2POp(dist.irecv, dst, k))
    works = dist.batch_isend_irecv(ops) if ops else []

    def finish():
        for wk in works:
            wk.wait()
        for s, b, e, buf in landed:
            out3[s, b:e].copy_(buf)
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
