"""How fast can a fresh NumPy array be page-locked (dexct_host_pin) by 1 .. 16 threads?  6.5 GB as the benchmark's result.
    gpurun -- python tools/probes/pin_threads.py"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dex_ct_sim_amd import _native

lib = _native.load()
torch.zeros(1, device='cuda')
n_bytes = 6_553_600_000
for threads in (1, 2, 4, 8, 16):
    a = np.empty(n_bytes, dtype=np.uint8)
    base = -(-a.ctypes.data // 4096) * 4096
    per = (n_bytes - 4096) // threads // (1 << 21) * (1 << 21)
    spans = [(base + k * per, per) for k in range(threads)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as pool:
        rcs = list(pool.map(lambda sp: lib.dexct_host_pin(sp[0], sp[1], 0), spans))
    t1 = time.perf_counter()
    with ThreadPoolExecutor(threads) as pool:
        list(pool.map(lambda sp: lib.dexct_host_unpin(sp[0], 0), spans))
    t2 = time.perf_counter()
    print(f'{threads:2d} threads: lock {per * threads / 1e9:.2f} GB in {t1 - t0:.3f} s = {per * threads / (t1 - t0) / 1e9:.1f} GB/s (return codes {set(rcs)}); unlock {t2 - t1:.3f} s', flush=True)
    del a
t0 = time.perf_counter()
h = torch.empty(n_bytes, dtype=torch.uint8, pin_memory=True)
print(f'torch.empty(pin_memory=True) of the same size: {time.perf_counter() - t0:.3f} s')
