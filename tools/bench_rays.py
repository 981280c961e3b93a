"""A/B of rays_kernel's slab batching (DEXCT_RAYS_BATCH = 1: the serial slab loop in 256-thread blocks; 4 / 8 / 16: one
wave per block, that many slabs with all their byte loads in flight) on the reference's own single-row scans.  Path
lengths and counts are compared bit for bit with batch 1."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import forward_project as fp, synthetic

det = os.path.join(ROOT, 'dex-ct-sim_amd/input/detector/eta_eid_mv.bin')
specs = [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)]
print('| scan | batch | ms | rays/s | bit-identical to batch 1 |')
print('|---|---|---|---|---|')
for name, n, views, chans in (('512^2, 1000 x 800', 512, 1000, 800), ('512^2, 1200 x 800 (params.txt)', 512, 1200, 800),
                              ('1024^2, 2000 x 1024', 1024, 2000, 1024), ('256^2, 360 x 512', 256, 360, 512)):
    ph = synthetic.make_phantom(n, 1, extent=51.2, seed=1234)
    ct = dx.FanBeamGeometry(N_channels=chans, N_proj=views, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=True,
                            detector_file=det, N_rows=1)
    pj = fp.Projector(ct, ph, kernel=1)
    _, mu_d, w_d, _ = pj.upload_tables(specs)
    ref = None
    for batch in ('1', '4', '8', '16'):
        os.environ['DEXCT_RAYS_BATCH'] = batch
        c, pl = pj.project_tables(mu_d, w_d, want_pathlen=True, layout=0)
        if ref is None:
            ref = (c.clone(), pl.clone())
        same = bool(torch.equal(c, ref[0]) and torch.equal(pl, ref[1]))
        out = torch.empty_like(c)
        pj.project_tables(mu_d, w_d, out=out, layout=0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            pj.project_tables(mu_d, w_d, out=out, layout=0)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f'| {name} | {batch} | {ms:.3f} | {views * chans / ms * 1e3:.3g} | {same} |', flush=True)
os.environ.pop('DEXCT_RAYS_BATCH', None)
