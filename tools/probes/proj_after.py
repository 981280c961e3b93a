"""Why does the projection take 11.6 ms inside the bench step and 10.3 ms back to back?  Times rows16_kernel right after
(a) another projection, (b) a 13 GB device-to-device copy (caches and TLBs flushed, no arithmetic), (c) the Newton kernel."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import forward_project as fp, matdecomp as md, synthetic

det = os.path.join(ROOT, 'dex-ct-sim_amd/input/detector/eta_eid_mv.bin')
n, views, chans = 512, 1000, 800
ct = dx.FanBeamGeometry(N_channels=chans, N_proj=views, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=True, detector_file=det, N_rows=n)
ph = synthetic.make_phantom(n, n, extent=51.2, seed=1234)
specs = [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)]
pj = fp.Projector(ct, ph)
_, mu_d, w_d, air = pj.upload_tables(specs)
out = torch.empty((2, views, chans, n), dtype=torch.float32, device='cuda')
log = torch.empty_like(out)
_, i0, mus = md.decomposition_tables(ct, specs[0], specs[1])
a = torch.empty((views, chans, n, 2), dtype=torch.float64, device='cuda')
big = torch.empty(13 * 2 ** 30 // 4, dtype=torch.float32, device='cuda')
big2 = torch.empty_like(big)
pj.project_tables(mu_d, w_d, out=out, layout=None, air=air, log_out=log)
gmax = out[0].max().double()


def timed_after(prep, reps=3):
    ts = []
    for _ in range(reps):
        prep()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        pj.project_tables(mu_d, w_d, out=out, layout=None, air=air, log_out=log)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return ' '.join(f'{t:.2f}' for t in ts)


print('after a projection      :', timed_after(lambda: pj.project_tables(mu_d, w_d, out=out, layout=None, air=air, log_out=log)))
print('after a 13 GB copy      :', timed_after(lambda: big2.copy_(big)))
print('after the Newton kernel :', timed_after(lambda: md.gn_device(out[0], out[1], i0, mus, 50, 'f64', out=a, mask_max=gmax)))
print('after Newton + 0.2 s idle:', timed_after(lambda: (md.gn_device(out[0], out[1], i0, mus, 50, 'f64', out=a, mask_max=gmax), torch.cuda.synchronize(), __import__('time').sleep(0.2))))
