"""rows16_kernel, round 4: detection with the air term folded into per-pair weight tables (DEXCT_P16_FOLD, default 1)
against the plain form (NOTE: the folded kernel lost and was not committed - profiles/r04_notes_rows16.md; with the
committed library both settings run the plain form), on the benchmark scan (512^3, 1000 x 800 x 512, dual spectrum) and on configs[1]'s
(256^3, 360 x 512 x 256, one spectrum): time, counts difference, log sinogram."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import forward_project as fp, synthetic

det = os.path.join(ROOT, 'dex-ct-sim_amd/input/detector/eta_eid_mv.bin')
for n, views, chans, specs in ((512, 1000, 800, [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)]),
                               (256, 360, 512, [synthetic.kramers_spectrum(120)])):
    ct = dx.FanBeamGeometry(N_channels=chans, N_proj=views, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=True, detector_file=det, N_rows=n)
    ph = synthetic.make_phantom(n, n, extent=51.2, seed=1234)
    pj = fp.Projector(ct, ph)
    _, mu_d, w_d, air = pj.upload_tables(specs)
    S = len(specs)
    out = {k: torch.empty((S, views, chans, n), dtype=torch.float32, device='cuda') for k in ('0', '1')}
    log = torch.empty((S, views, chans, n), dtype=torch.float32, device='cuda')
    times = {'0': [], '1': []}
    for rep in range(3):
        for fold in ('1', '0'):
            os.environ['DEXCT_P16_FOLD'] = fold
            pj.project_tables(mu_d, w_d, out=out[fold], layout=None, air=air, log_out=log)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                pj.project_tables(mu_d, w_d, out=out[fold], layout=None, air=air, log_out=log)
            e1.record()
            torch.cuda.synchronize()
            times[fold].append(e0.elapsed_time(e1) / 5)
    rel = float(((out['1'] - out['0']).abs() / out['0']).max())
    print(f'{n}^3 {views} x {chans} x {n}, {S} spectra: fold {min(times["1"]):.3f} ms, plain {min(times["0"]):.3f} ms, '
          f'counts max rel diff {rel:.2e}', flush=True)
os.environ.pop('DEXCT_P16_FOLD', None)
