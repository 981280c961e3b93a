"""The Newton kernel at the reference's own size (input/params.txt: 1200 views x 800 channels, one row = 9.6e5 pixels):
pixels per lane of a wave's run (DEXCT_GN_CHUNK) against time."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import forward_project as fp, matdecomp as md, synthetic

det = os.path.join(ROOT, 'dex-ct-sim_amd/input/detector/eta_eid_mv.bin')
for views, chans, n in ((1200, 800, 512), (360, 512, 256), (2000, 1024, 1024)):
    ct = dx.FanBeamGeometry(N_channels=chans, N_proj=views, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=True, detector_file=det, N_rows=1)
    ph = synthetic.make_phantom(n, 1, extent=51.2, seed=1234)
    specs = [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)]
    pj = fp.Projector(ct, ph)
    _, mu_d, w_d, _ = pj.upload_tables(specs)
    counts = pj.project_tables(mu_d, w_d, layout=None)
    _, i0, mus = md.decomposition_tables(ct, specs[0], specs[1])
    gmax = counts[0].max().double()
    a = torch.empty(tuple(counts[0].shape) + (2,), dtype=torch.float64, device='cuda')
    ref = None
    for chunk in ('', '1', '2', '3', '4', '6', '8', '12'):
        os.environ.pop('DEXCT_GN_CHUNK', None)
        if chunk:
            os.environ['DEXCT_GN_CHUNK'] = chunk
        md.gn_device(counts[0], counts[1], i0, mus, 50, 'f64', out=a, mask_max=gmax)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            md.gn_device(counts[0], counts[1], i0, mus, 50, 'f64', out=a, mask_max=gmax)
        e1.record()
        torch.cuda.synchronize()
        if ref is None:
            ref = a.clone()
        print(f'{views} x {chans}: chunk {chunk or "default"}: {e0.elapsed_time(e1) / 5:.3f} ms   bit-identical: {bool(torch.equal(a.view(torch.int64), ref.view(torch.int64)))}', flush=True)
    os.environ.pop('DEXCT_GN_CHUNK', None)
