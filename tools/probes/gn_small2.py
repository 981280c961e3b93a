"""The Newton kernel at the reference's own size (1200 views x 800 channels, one row = 9.6e5 pixels) and at configs[1]'s:
workgroups per CU of the run-queue grid (DEXCT_GN_BLOCKS_PER_CU) x pixels per lane and fetch (DEXCT_GN_CHUNK)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import forward_project as fp, matdecomp as md, synthetic

det = os.path.join(ROOT, 'dex-ct-sim_amd/input/detector/eta_eid_mv.bin')
for views, chans, n, rows in ((1200, 800, 512, 1), (360, 512, 256, 1), (100, 800, 512, 64), (1000, 800, 512, 32)):
    ct = dx.FanBeamGeometry(N_channels=chans, N_proj=views, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=True, detector_file=det, N_rows=rows)
    ph = synthetic.make_phantom(n, rows, extent=51.2, seed=1234)
    specs = [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)]
    pj = fp.Projector(ct, ph)
    _, mu_d, w_d, _ = pj.upload_tables(specs)
    counts = pj.project_tables(mu_d, w_d, layout=None)
    _, i0, mus = md.decomposition_tables(ct, specs[0], specs[1])
    gmax = counts[0].max().double()
    a = torch.empty(tuple(counts[0].shape) + (2,), dtype=torch.float64, device='cuda')
    ref = None
    for cap in ('', '1', '2', '3', '4', '5', '8'):
        for chunk in ('1', '2'):
            for k in ('DEXCT_GN_CHUNK', 'DEXCT_GN_BLOCKS_PER_CU'):
                os.environ.pop(k, None)
            if cap:
                os.environ['DEXCT_GN_BLOCKS_PER_CU'] = cap
                os.environ['DEXCT_GN_CHUNK'] = chunk
            elif chunk == '2':
                continue
            md.gn_device(counts[0], counts[1], i0, mus, 50, 'f64', out=a, mask_max=gmax)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                md.gn_device(counts[0], counts[1], i0, mus, 50, 'f64', out=a, mask_max=gmax)
            e1.record()
            torch.cuda.synchronize()
            if ref is None:
                ref = a.clone()
            print(f'{views} x {chans} x {rows} = {a.numel() // 2:.3g} pixels: blocks/CU {cap or "default"} chunk {chunk if cap else "default"}: '
                  f'{e0.elapsed_time(e1) / 5:.3f} ms   bit-identical: {bool(torch.equal(a.view(torch.int64), ref.view(torch.int64)))}', flush=True)
for k in ('DEXCT_GN_CHUNK', 'DEXCT_GN_BLOCKS_PER_CU'):
    os.environ.pop(k, None)
