"""(separate process: the probe above may leave the runtime in an error state)  Lock memory that was only touched (written) by
the CPU first; and the rate of the touch itself."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dex_ct_sim_amd import _native

lib = _native.load()
torch.zeros(1, device='cuda')
n_bytes = 3_276_800_000
a = np.empty(n_bytes + 8192, dtype=np.uint8)
base = -(-a.ctypes.data // 4096) * 4096
n = n_bytes // 4096 * 4096
t0 = time.perf_counter()
a[::4096] = 1
t1 = time.perf_counter()
print(f'touch every page: {t1 - t0:.3f} s = {n_bytes / (t1 - t0) / 1e9:.1f} GB/s')
rc = lib.dexct_host_pin(base, n, 0)
t2 = time.perf_counter()
print(f'lock touched memory, one span: {t2 - t1:.3f} s = {n_bytes / (t2 - t1) / 1e9:.1f} GB/s (rc {rc})')
d = torch.empty(n, dtype=torch.uint8, device='cuda')
t = torch.from_numpy(a[base - a.ctypes.data:][:n])
for k in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t.copy_(d, non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'torch download into it: {dt:.3f} s = {n / dt / 1e9:.1f} GB/s')
h = torch.empty(n, dtype=torch.uint8, pin_memory=True)
for k in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    h.copy_(d, non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'torch download into a torch page-locked tensor: {dt:.3f} s = {n / dt / 1e9:.1f} GB/s')
