import os, sys, time
ROOT='/root/repo'
sys.path.insert(0, ROOT)
import numpy as np, torch
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import synthetic
det = os.path.join(ROOT, 'dex-ct-sim_amd/input/detector/eta_eid_mv.bin')
ct = dx.FanBeamGeometry(N_channels=800, N_proj=1200, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=True, detector_file=det, N_rows=1)
ph = synthetic.make_phantom(512, 1, extent=51.2, seed=1234)
s1, s2 = synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)
def once():
    t0=time.perf_counter()
    r1,l1 = dx.get_sino(ct, ph, s1)
    t1=time.perf_counter()
    r2,l2 = dx.get_sino(ct, ph, s2)
    t2=time.perf_counter()
    a,b = dx.get_basismat_sinos(ct, r1, r2, s1, s2, n_iters=50)
    t3=time.perf_counter()
    return (t1-t0)*1e3,(t2-t1)*1e3,(t3-t2)*1e3
for k in range(6):
    print('get_sino %.2f ms, get_sino %.2f ms, get_basismat_sinos %.2f ms' % once(), flush=True)
import cProfile, pstats
pr=cProfile.Profile(); pr.enable()
for k in range(10): once()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
