"""dexct_host_touch: page-fault rate of fresh memory by 1 .. 16 threads, and the lock that follows (free once resident).
    gpurun -- python tools/probes/touch_threads.py"""
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
n_bytes = 6_553_600_000
print('cpus of this process:', len(os.sched_getaffinity(0)))
for threads in (1, 2, 4, 8, 16, 32):
    a = np.empty(n_bytes + 8192, dtype=np.uint8)
    base = -(-a.ctypes.data // 4096) * 4096
    n = n_bytes // 4096 * 4096
    t0 = time.perf_counter()
    rc = lib.dexct_host_touch(base, n, threads)
    t1 = time.perf_counter()
    rc2 = lib.dexct_host_pin(base, n, 0)
    t2 = time.perf_counter()
    lib.dexct_host_unpin(base, 0)
    t3 = time.perf_counter()
    print(f'{threads:2d} threads: touch {t1 - t0:.3f} s = {n / (t1 - t0) / 1e9:.1f} GB/s (rc {rc}); lock {t2 - t1:.4f} s (rc {rc2}); unlock {t3 - t2:.4f} s', flush=True)
    del a
