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
