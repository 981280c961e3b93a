"""get_basismat_sinos through the NumPy boundary at benchmark size: plain sequence against the pipelined one."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import forward_project as fp, matdecomp as md, synthetic

det = os.path.join(ROOT, 'dex-ct-sim_amd/input/detector/eta_eid_mv.bin')
n = 512
ct = dx.FanBeamGeometry(N_channels=800, N_proj=1000, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=True, detector_file=det, N_rows=n)
ph = synthetic.make_phantom(n, n, extent=51.2, seed=1234)
specs = [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)]
r1, _ = dx.get_sino(ct, ph, specs[0])
r2, _ = dx.get_sino(ct, ph, specs[1])
print('inputs pinned:', torch.from_numpy(r1).is_pinned(), r1.shape, r1.dtype)
for label, chunks, minpix in (('pipelined 2', 2, 1), ('pipelined 3', 3, 1), ('pipelined 4', 4, 1), ('pipelined 5', 5, 1), ('pipelined 6', 6, 1), ('pipelined 8', 8, 1), ('pipelined 12', 12, 1), ('pipelined 16', 16, 1), ('plain', 8, 1 << 62)):
    md._PIPE_CHUNKS, md._PIPE_MIN_PIXELS = chunks, minpix
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m1, m2 = dx.get_basismat_sinos(ct, r1, r2, specs[0], specs[1], n_iters=50)
        dt = time.perf_counter() - t0
        chk = float(m1[500, 256, 400])
        del m1, m2
        print(f'{label}: {dt:.3f} s   (sample {chk:.12g})', flush=True)
