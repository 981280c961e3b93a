"""get_basismat_sinos through the NumPy boundary at benchmark size, eight calls in a row: the spread, with and without locking
the inputs, and with the results kept alive (new blocks) or dropped (pooled blocks).   gpurun -- python tools/probes/boundary_gn3.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import _device, matdecomp as md, synthetic

det = os.path.join(ROOT, 'dex-ct-sim_amd/input/detector/eta_eid_mv.bin')
n = 512
ct = dx.FanBeamGeometry(N_channels=800, N_proj=1000, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=True, detector_file=det, N_rows=n)
ph = synthetic.make_phantom(n, n, extent=51.2, seed=1234)
specs = [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)]
r1, _ = dx.get_sino(ct, ph, specs[0])
r2, _ = dx.get_sino(ct, ph, specs[1])


def calls(label, n_calls=6, keep=False):
    held = []
    for rep in range(n_calls):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m1, m2 = dx.get_basismat_sinos(ct, r1, r2, specs[0], specs[1], n_iters=50)
        dt = time.perf_counter() - t0
        if keep:
            held.append((m1, m2))
        del m1, m2
        print(f'{label}: {dt:.3f} s', flush=True)


calls('default')
real = _device.locked_arrays
md.locked_arrays = lambda lib, arrs, d: real(lib, arrs, d, min_bytes=1 << 62)
calls('inputs not locked (pageable uploads)', 3)
md.locked_arrays = real
p1, p2 = torch.from_numpy(r1).pin_memory().numpy(), torch.from_numpy(r2).pin_memory().numpy()
r1, r2 = p1, p2
calls('inputs in torch page-locked memory', 4)
for chunks in (4, 16):
    md._PIPE_CHUNKS = chunks
    calls(f'{chunks} chunks', 3)
