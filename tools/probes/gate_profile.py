"""Where the calibration of the short cut's table (once per pair of spectra, then from DEXCT_CACHE_DIR) spends its time: cProfile of
one matdecomp.calibrate_gate for the benchmark's Kramers pair."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import matdecomp as md, synthetic
from dex_ct_sim_amd._device import to_dev

os.environ['DEXCT_CACHE_DIR'] = 'off'
dev = torch.device('cuda:0')
det = os.path.join(ROOT, 'dex-ct-sim_amd/input/detector/eta_eid_mv.bin')
ct = dx.FanBeamGeometry(N_channels=800, N_proj=1200, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=True, detector_file=det, N_rows=1)
_, i0, mus = md.decomposition_tables(ct, synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80))
i0_d, mus_d = to_dev(i0, torch.float64, dev)[:, None, :].contiguous(), to_dev(mus, torch.float64, dev)
args = (np.ascontiguousarray(i0), np.ascontiguousarray(mus), i0_d, mus_d, dev, 1e-12)
md.calibrate_gate(*args)
t0 = time.perf_counter()
md.calibrate_gate(*args)
print(f'calibrate_gate {time.perf_counter() - t0:.3f} s')
pr = cProfile.Profile()
pr.enable()
md.calibrate_gate(*args)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(30)
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
