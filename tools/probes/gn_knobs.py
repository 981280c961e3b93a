"""The default (one-step) Newton launch on the benchmark's sinograms against its run-time knobs: workgroups per CU
(dexct_gn_options.blocks_per_cu) and the order of the tile hand-out.   gpurun -- python tools/probes/gn_knobs.py"""
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
_, mu_d, w_d, _ = pj.upload_tables(specs)
counts = pj.project_tables(mu_d, w_d, layout=None)
_, i0, mus = md.decomposition_tables(ct, specs[0], specs[1])
gmax = counts[0].max().double()
a = torch.empty((views, n, chans, 2), dtype=torch.float64, device=counts.device)


def run(reps=6, **kw):
    md.gn_device(counts[0], counts[1], i0, mus, 50, 'f64', out=a, mask_max=gmax, out_rc=(n, chans), **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        md.gn_device(counts[0], counts[1], i0, mus, 50, 'f64', out=a, mask_max=gmax, out_rc=(n, chans), **kw)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ref = None
for rep in range(2):
    for kw in ({}, {'blocks_per_cu': 3}, {'blocks_per_cu': 2}, {'blocks_per_cu': 1}, {'natural_order': True}):
        ms = run(**kw)
        if ref is None:
            ref = a.clone()
        same = bool(torch.equal(a.view(torch.int64), ref.view(torch.int64)))
        print(f'{kw or "default"}: {ms:.2f} ms  bit-identical: {same}  {md.last_gn_stats().get("mode")}', flush=True)
