"""A/B of the three lane mappings of the traversal on one MI355X:
  kernel 1  rays_kernel      lanes = adjacent channels (one thread per ray)
  kernel 3  rows4_kernel     lanes = detector rows, 4 rows per lane (stacked fans only; shares the in-plane geometry)
  kernel 6  wave_ray_kernel  lanes = dominant-axis slabs of ONE ray (the north star's "one wavefront per ray")
on (a) the reference's own single-row 2-D scan and (b) a stacked fan of the same geometry.  All produce the same
per-material path lengths bit for bit (checked here).  Prints one markdown table row per line."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import forward_project as fp, synthetic

det = os.path.join(ROOT, 'dex-ct-sim_amd/input/detector/eta_eid_mv.bin')
specs = [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)]


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print('| scan | kernel | ms | rays/s | path lengths identical to kernel 1 |')
print('|---|---|---|---|---|')
for name, n, nz, views, chans, rows, kernels in (
        ('single row 512^2, 1000 x 800 (the reference\'s own scan)', 512, 1, 1000, 800, 1, (1, 6)),
        ('single row 1024^2, 2000 x 1024', 1024, 1, 2000, 1024, 1, (1, 6)),
        ('stacked fan 512^3, 100 x 800 x 512 rows', 512, 512, 100, 800, 512, (1, 3, 6)),
):
    ph = synthetic.make_phantom(n, nz, extent=51.2, seed=1234)
    ct = dx.FanBeamGeometry(N_channels=chans, N_proj=views, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=True,
                            detector_file=det, N_rows=rows)
    ref = None
    for k in kernels:
        pj = fp.Projector(ct, ph, kernel=k)
        _, mu_d, w_d, _ = pj.upload_tables(specs)
        c, pl = pj.project_tables(mu_d, w_d, want_pathlen=True, layout=0)
        if ref is None:
            ref = pl
        same = bool(torch.equal(pl, ref))
        out = torch.empty_like(c)
        ms = timed(lambda: pj.project_tables(mu_d, w_d, out=out, layout=0))
        nr = c[0].numel()
        print(f'| {name} | {k} | {ms:.3f} | {nr / ms * 1e3:.3g} | {same} |', flush=True)
        del pj, c, pl, out
