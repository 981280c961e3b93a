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


def gather_views(local, n_views, view_dim=0, async_op=False):
    """All-gather view shards (possibly of unequal size) into the full tensor on every rank.

    With ``async_op=True`` returns ``finish`` - call it to wait and get the tensor (lets the caller
    overlap the collective with compute; on the nccl/RCCL backend the transfer runs on the process
    group's own stream)."""
    r, w = world()
    if w == 1:
        return (lambda: local) if async_op else local
    local = local.movedim(view_dim, 0).contiguous()
    sizes = [split(n_views, k, w) for k in range(w)]
    n_max = max(e - b for b, e in sizes)
    pad = n_max - local.shape[0]
    if pad:
        local = torch.cat([local, local.new_zeros((pad,) + tuple(local.shape[1:]))], dim=0)
    dev = local.device
    staged = _needs_cpu_staging(local)
    src = local.cpu() if staged else local
    out = src.new_empty((w * n_max,) + tuple(src.shape[1:]))
    work = dist.all_gather_into_tensor(out, src, async_op=async_op)

    def finish():
        if async_op:
            work.wait()
        parts = [out[k * n_max:k * n_max + (e - b)] for k, (b, e) in enumerate(sizes)]
        full = torch.cat(parts, dim=0) if pad or w * n_max != n_views else out
        return full.to(dev).movedim(0, view_dim)

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
