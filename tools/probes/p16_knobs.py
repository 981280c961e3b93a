"""rows16_kernel on the benchmark scan without the log output (what the step runs since dexct_transpose_log): the register
allocation (DEXCT_P16_MINW), the view tile of the block mapping (DEXCT_VIEW_TILE) and the per-energy masks (DEXCT_DET_MASKS) once
more.   gpurun -- python tools/probes/p16_knobs.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import forward_project as fp, synthetic

det = os.path.join(ROOT, 'dex-ct-sim_amd/input/detector/eta_eid_mv.bin')
n, views, chans = 512, 1000, 800
ct = dx.FanBeamGeometry(N_channels=chans, N_proj=views, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=True, detector_file=det, N_rows=n)
ph = synthetic.make_phantom(n, n, extent=51.2, seed=1234)
specs = [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)]
pj = fp.Projector(ct, ph)
_, mu_d, w_d, air = pj.upload_tables(specs)
out = torch.empty((2, views, chans, n), dtype=torch.float32, device='cuda')
ref = None


def run(env, reps=5):
    for k in ('DEXCT_P16_MINW', 'DEXCT_VIEW_TILE', 'DEXCT_DET_MASKS'):
        os.environ.pop(k, None)
    os.environ.update(env)
    pj.project_tables(mu_d, w_d, out=out, layout=None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        pj.project_tables(mu_d, w_d, out=out, layout=None)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for rep in range(2):
    for env in ({}, {'DEXCT_P16_MINW': '3'}, {'DEXCT_P16_MINW': '5'}, {'DEXCT_P16_MINW': '6'}, {'DEXCT_P16_MINW': '8'}, {'DEXCT_VIEW_TILE': '4'},
                {'DEXCT_VIEW_TILE': '16'}, {'DEXCT_VIEW_TILE': '32'}, {'DEXCT_DET_MASKS': '0'}):
        ms = run(env)
        if ref is None:
            ref = out.clone()
        print(f'{env or "default"}: {ms:.3f} ms   bit-identical: {bool(torch.equal(out, ref))}', flush=True)
