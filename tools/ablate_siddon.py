import sys, os, torch, numpy as np
sys.path.insert(0, os.getcwd())
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import forward_project as fp, synthetic
det = 'dex-ct-sim_amd/input/detector/eta_eid_mv.bin'
ct = dx.FanBeamGeometry(800, 1000, detector_file=det, N_rows=512)
ph = synthetic.make_phantom(512, 512)
pj = fp.Projector(ct, ph)
def timeit(mu, w, n=3):
    out = torch.empty((w.shape[0], 1000, 512, 800), dtype=torch.float32, device='cuda')
    pj.project_tables(mu, w, out=out); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): pj.project_tables(mu, w, out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
for nE, S in ((134, 2), (134, 1), (64, 2), (2, 2), (2, 1)):
    mu = torch.rand((3, nE), device='cuda') * 0.2
    w = torch.rand((S, nE), device='cuda')
    print('nE', nE, 'S', S, 'ms', round(timeit(mu, w), 2))
# all-air volume: no corrections, same loads
ph2 = synthetic.make_phantom(512, 512); ph2.volume[:] = 0
pj = fp.Projector(ct, ph2)
mu = torch.rand((3, 2), device='cuda') * 0.2; w = torch.rand((1, 2), device='cuda')
print('air volume nE 2:', round(timeit(mu, w), 2))
