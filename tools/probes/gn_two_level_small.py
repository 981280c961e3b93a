"""Two-level Newton modes against the single launch at small and medium sizes (kernel time of the whole gn_device call,
best of 5): where 'coarse' (two launches) overtakes 'start' (one).  gpurun -- python tools/probes/gn_two_level_small.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import forward_project as fp, matdecomp as md, synthetic
from dex_ct_sim_amd._device import ptr, stream_ptr

det = os.path.join(ROOT, 'dex-ct-sim_amd', 'input', 'detector', 'eta_eid_mv.bin')
specs = [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)]
for views, chans, rows, n in ((120, 128, 1, 256), (360, 256, 1, 256), (1200, 800, 1, 512), (2000, 1024, 1, 512), (200, 400, 32, 256),
                              (360, 512, 64, 256), (360, 512, 256, 256), (250, 800, 512, 512)):
    ct = dx.FanBeamGeometry(N_channels=chans, N_proj=views, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=True, detector_file=det, N_rows=rows)
    ph = synthetic.make_phantom(n, rows, extent=51.2, seed=1234)
    pj = fp.Projector(ct, ph)
    _, mu_d, w_d, air = pj.upload_tables(specs)
    counts = pj.project_tables(mu_d, w_d, layout=None)
    _, i0, mus = md.decomposition_tables(ct, specs[0], specs[1])
    gmax = torch.empty((), dtype=torch.float64, device='cuda')
    pj.lib.dexct_reduce_max(ptr(counts[0]), 0, counts[0].numel(), ptr(gmax), stream_ptr())
    res, ref = {}, None
    for mode, kw in (('exact', dict(stop_tol=0.0, two_level=False)), ('single', dict(two_level=False)), ('start', dict(two_level='start')),
                     ('coarse', dict(two_level='coarse'))):
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            a = md.gn_device(counts[0], counts[1], i0, mus, 50, 'f64', mask_max=gmax, mask_frac=0.95, **kw)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        st = md.last_gn_stats()
        if mode == 'exact':
            ref = a
            dev = 0.0
        else:
            assert torch.equal(torch.isnan(a), torch.isnan(ref))
            dev = float(torch.nan_to_num((a - ref).abs() / ref.abs().clamp(min=1.0), nan=0.0).max())
        res[mode] = (best, dev, st['pixel_iterations'], st.get('coarse_pixel_iterations', 0))
    print(f'{views} x {chans} x {rows} = {counts[0].numel():.3g} pixels: ' +
          ', '.join(f'{m} {v[0]:.3f} ms (dev {v[1]:.1e}; full steps {v[2]:.3g}, coarse {v[3]:.3g})' for m, v in res.items()), flush=True)
