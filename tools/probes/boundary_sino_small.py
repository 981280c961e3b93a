"""Where get_sino spends its time at the reference's own size (1200 x 800, one row): cProfile of 200 warm calls."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import synthetic

det = os.path.join(ROOT, 'dex-ct-sim_amd/input/detector/eta_eid_mv.bin')
ct = dx.FanBeamGeometry(N_channels=800, N_proj=1200, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=True, detector_file=det, N_rows=1)
ph = synthetic.make_phantom(512, 1, extent=51.2, seed=1234)
s1 = synthetic.kramers_spectrum(140)
for _ in range(5):
    dx.get_sino(ct, ph, s1)
t0 = time.perf_counter()
for _ in range(200):
    dx.get_sino(ct, ph, s1)
print(f'get_sino warm: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms per call')
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    dx.get_sino(ct, ph, s1)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(28)
