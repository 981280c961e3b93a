"""A/B of rows16_kernel's output side on the benchmark scan (512^3, 1000 x 800 x 512, dual spectrum): whole-line stores
through LDS (DEXCT_P16_STAGED, default 1) against the per-round 16-byte stores, with and without the log sinogram.
Results are compared bit for bit.   python tools/bench_p16.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import forward_project as fp, synthetic

det = os.path.join(ROOT, 'dex-ct-sim_amd/input/detector/eta_eid_mv.bin')
n, views, chans = int(os.environ.get('N', 512)), int(os.environ.get('VIEWS', 1000)), int(os.environ.get('CHANS', 800))
ct = dx.FanBeamGeometry(N_channels=chans, N_proj=views, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=True,
                        detector_file=det, N_rows=n)
ph = synthetic.make_phantom(n, n, extent=51.2, seed=1234)
specs = [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)]
pj = fp.Projector(ct, ph)
_, mu_d, w_d, air = pj.upload_tables(specs)
shape = (2, views, chans, n)
out = torch.empty(shape, dtype=torch.float32, device='cuda')
log = torch.empty(shape, dtype=torch.float32, device='cuda')
ref = None


def run(staged, want_log, reps=5, minw='4'):
    os.environ['DEXCT_P16_STAGED'] = staged
    os.environ['DEXCT_P16_MINW'] = minw
    kw = dict(air=air, log_out=log) if want_log else {}
    pj.project_tables(mu_d, w_d, out=out, layout=None, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        pj.project_tables(mu_d, w_d, out=out, layout=None, **kw)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for rep in range(2):
    for staged in ('1', '0'):
        for want_log in (False, True):
            ms = run(staged, want_log)
            if ref is None:
                ref = out.clone()
            same = bool(torch.equal(out, ref))
            print(f'staged={staged} log={int(want_log)}: {ms:.3f} ms   counts bit-identical: {same}', flush=True)
    ms = run('1', True, minw='5')
    print(f'staged=1 log=1 minw=5: {ms:.3f} ms   counts bit-identical: {bool(torch.equal(out, ref))}', flush=True)
