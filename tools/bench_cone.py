"""Cone-beam kernel at benchmark scale: 512^3, 1000 x 800 x 512 rays, both spectra."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import forward_project as fp, synthetic

det = os.path.join(ROOT, 'dex-ct-sim_amd/input/detector/eta_eid_mv.bin')
views = int(sys.argv[1]) if len(sys.argv) > 1 else 250
ct = dx.FanBeamGeometry(800, views, detector_file=det, N_rows=512, cone=True, h_iso=0.1)
ph = synthetic.make_phantom(512, 512)
specs = [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)]
pj = fp.Projector(ct, ph)
_, mu_d, w_d, _ = pj.upload_tables(specs)
out = pj.project_tables(mu_d, w_d)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(2):
    pj.project_tables(mu_d, w_d, out=out)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 2
rays = views * 512 * 800
print(f'cone kernel: {views} views x 512 rows x 800 ch = {rays:.3g} rays in {ms:.1f} ms = {rays / ms * 1e3:.3g} rays/s')
