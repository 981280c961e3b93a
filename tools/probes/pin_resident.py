"""Page-locking memory that is RESIDENT already (touched, or locked and unlocked before) against fresh memory; one span against
several; a torch copy from a block registered in several spans.   gpurun -- python tools/probes/pin_resident.py"""
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


def block():
    a = np.empty(n_bytes + 8192, dtype=np.uint8)
    return a, -(-a.ctypes.data // 4096) * 4096, n_bytes // 4096 * 4096


def timed(label, fn):
    t0 = time.perf_counter()
    rc = fn()
    dt = time.perf_counter() - t0
    print(f'{label}: {dt:.3f} s = {n_bytes / dt / 1e9:.1f} GB/s (rc {rc})', flush=True)


a, base, n = block()
timed('fresh memory, one span: lock', lambda: lib.dexct_host_pin(base, n, 0))
timed('  unlock', lambda: lib.dexct_host_unpin(base, 0))
timed('locked before (resident), one span: lock', lambda: lib.dexct_host_pin(base, n, 0))
timed('  unlock', lambda: lib.dexct_host_unpin(base, 0))
half = n // 2 // 4096 * 4096
timed('resident, two spans: lock', lambda: (lib.dexct_host_pin(base, half, 0), lib.dexct_host_pin(base + half, n - half, 0)))
d = torch.empty(n_bytes, dtype=torch.uint8, device='cuda')
t = torch.from_numpy(a[base - a.ctypes.data:][:n])
print('torch sees it as page-locked:', t.is_pinned())
for label, sl in (('inside span 1', slice(0, half)), ('inside span 2', slice(half, n)), ('across both', slice(0, n))):
    try:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        d[sl].copy_(t[sl], non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f'torch upload {label}: {dt:.3f} s = {(sl.stop - sl.start) / dt / 1e9:.1f} GB/s')
    except Exception as e:
        print(f'torch upload {label}: {type(e).__name__}: {str(e).splitlines()[0]}')
        break
