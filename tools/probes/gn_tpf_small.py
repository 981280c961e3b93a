"""Tiles reserved per atomic (DEXCT_GN_TILES_PER_FETCH) at small and medium sizes, short-cut launch, best of 7.
gpurun -- python tools/probes/gn_tpf_small.py"""
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
for views, chans, rows, n in ((360, 256, 1, 256), (1200, 800, 1, 512), (2000, 1024, 1, 512), (200, 400, 32, 256), (360, 512, 64, 256)):
    ct = dx.FanBeamGeometry(N_channels=chans, N_proj=views, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=True, detector_file=det, N_rows=rows)
    ph = synthetic.make_phantom(n, rows, extent=51.2, seed=1234)
    pj = fp.Projector(ct, ph)
    _, mu_d, w_d, air = pj.upload_tables(specs)
    counts = pj.project_tables(mu_d, w_d, layout=None)
    _, i0, mus = md.decomposition_tables(ct, specs[0], specs[1])
    gmax = torch.empty((), dtype=torch.float64, device='cuda')
    pj.lib.dexct_reduce_max(ptr(counts[0]), 0, counts[0].numel(), ptr(gmax), stream_ptr())
    out = []
    for tpf in (0, 1, 2, 4, 8):
        if tpf:
            os.environ['DEXCT_GN_TILES_PER_FETCH'] = str(tpf)
        else:
            os.environ.pop('DEXCT_GN_TILES_PER_FETCH', None)
        best = 1e9
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            md.gn_device(counts[0], counts[1], i0, mus, 50, 'f64', mask_max=gmax, mask_frac=0.95)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        out.append(f'{"default" if not tpf else tpf}: {best:.3f} ms')
    print(f'{views} x {chans} x {rows} = {counts[0].numel():.3g} pixels ({-(-counts[0].numel() // 64)} tiles): ' + ', '.join(out), flush=True)
