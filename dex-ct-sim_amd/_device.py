"""torch-ROCm as the device container: allocation, streams, pointers.  No compute happens here."""
import os

import numpy as np
import torch

from ._native import DexctError


def device():
    if not torch.cuda.is_available():
        raise DexctError('no HIP device visible: the dex-ct hot path runs on an MI355X only (no CPU fallback)')
    idx = int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count()
    torch.cuda.set_device(idx)
    return torch.device('cuda', idx)


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return None if t is None else t.data_ptr()


def to_dev(a, dtype, dev):
    """NumPy array or torch tensor -> contiguous device tensor of dtype."""
    if isinstance(a, torch.Tensor):
        return a.to(device=dev, dtype=dtype).contiguous()
    return torch.from_numpy(np.ascontiguousarray(a)).to(device=dev, dtype=dtype)


def to_host(t):
    """Device tensor -> NumPy array through PAGE-LOCKED host memory (one DMA at PCIe rate instead of a staged copy
    through pageable memory).  The pinned buffer comes from torch's caching host allocator and is referenced by the
    returned array alone: the caller owns it like any fresh array, and the allocator reuses the pages once the array
    is garbage, so repeated calls do not pin new memory."""
    if not t.is_cuda:
        return t.numpy()
    host = pinned_empty(t.shape, t.dtype)
    host.copy_(t, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    return host.numpy()


def pinned_empty(shape, dtype):
    """Page-locked host tensor; pageable if the host refuses to lock that much memory (the copy is then staged by the
    runtime: slower, same result)."""
    try:
        return torch.empty(tuple(shape), dtype=dtype, pin_memory=True)
    except RuntimeError:
        return torch.empty(tuple(shape), dtype=dtype)
